/*
 * tinyimgcodec_hip.h - C-ABI of the MI355X-native tinyimgcodec hot path (libtinyimgcodec_hip.so).
 *
 * The reference (clysto/tinyimgcodec) has no FFI of its own: its boundary is the Python function API
 * tinyimgcodec/__init__.py:1-5 -> codec.py (encode/decode/compress/decompress).  Each entry point below names
 * the reference function (file:line under /root/reference) whose work it replaces.  Conventions: plain pointers
 * and sizes, caller-owned buffers, int return codes (0 = ok, negative = error, message via tic_last_error),
 * no exceptions across the ABI, no torch types.  A tic_ctx owns one HIP device + one stream; use one context per
 * host thread (calls on the same context must not overlap).  All device work is issued on the context's own
 * stream.  There is NO CPU fallback for the transform stage: without a usable gfx950 device tic_create fails.
 *
 * Coefficient layout produced by the device stage ("zz16"): int16 [N][64], N = ceil(h/8)*ceil(w/8) blocks in
 * raster order, each block's 64 quantised coefficients in zig-zag scan order (constants.py:23-34); element 0 is
 * the quantised DC *before* DPCM (the DPCM of codec.py:34-35 is applied by the entropy stage / tic_encode).
 */
#ifndef TINYIMGCODEC_HIP_H
#define TINYIMGCODEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIC_OK 0
#define TIC_E_ARG -1      /* bad argument (null pointer, negative size, ...) */
#define TIC_E_QUALITY -2  /* quality outside 1..99 (reference: ZeroDivisionError / KeyError / struct.error) */
#define TIC_E_RANGE -3    /* a coefficient has no Huffman code (reference: KeyError in encode_huffman) */
#define TIC_E_SPACE -4    /* output buffer too small */
#define TIC_E_STREAM -5   /* malformed / unsupported stream */
#define TIC_E_HIP -6      /* HIP runtime error (see tic_last_error) */
#define TIC_E_NODEVICE -7 /* no gfx950 device / extension unusable */
#define TIC_E_BUSY -8     /* an asynchronous call has not finished yet (tic_async_result with wait == 0) */

/* kernel variants of the transform stage (all bit-identical in output) */
#define TIC_QUALITY_CUSTOM 0 /* the quality installed with tic_set_custom_quality (any number in [1, 99]) */
#define TIC_KERNEL_AUTO 0
#define TIC_KERNEL_EXACT 1  /* every coefficient in pocketfft float64 operation order */
#define TIC_KERNEL_HYBRID 2 /* fp32 AAN fast path + guard band + exact rational coefficients + exact fallback */

typedef struct tic_ctx tic_ctx;

/* ---- lifecycle -------------------------------------------------------------------------------------- */
const char *tic_version(void);
/* 0 for the shipped library (it reads no environment variable); 1 for the test-hooks build of the same sources
 * (libtinyimgcodec_hip_hooks.so, -DTIC_TEST_HOOKS: csrc/tic_hooks.h lists the variables that build honours). */
int tic_build_has_test_hooks(void);
int tic_device_count(void);
/* Create a context on HIP device `device`.  NULL on failure (tic_last_error(NULL) explains). */
tic_ctx *tic_create(int device);
void tic_destroy(tic_ctx *ctx);
/* Last error message of the context (or of the failed tic_create when ctx == NULL).  Never NULL. */
const char *tic_last_error(const tic_ctx *ctx);
/* Device name / arch string of the context's device, e.g. "gfx950:sramecc+:xnack-". */
const char *tic_device_arch(const tic_ctx *ctx);

/* ---- sizes ------------------------------------------------------------------------------------------ */
/* Number of 8x8 blocks of an h x w image after pad_image (utils.py:56-61); 0 when h == 0 or w == 0. */
size_t tic_num_blocks(int h, int w);
/* Upper bound of the compressed size in bytes (16-byte header + worst-case Huffman payload). */
size_t tic_compress_bound(int h, int w);

/* ---- transform stage: replaces encode() codec.py:26-43 (pad_image utils.py:56-61, level shift codec.py:29,
 *      block_slice utils.py:13-20, block_dct utils.py:32-37, block_quantize utils.py:48-53, zig-zag
 *      codec.py:32-33).  Runs on the GPU. ---------------------------------------------------------------- */

/* Host-buffer convenience form: H2D copy, kernel, D2H copy, synchronous.  image: uint8, row stride in bytes.
 * coeffs_zz: int16[N*64] (layout above). */
int tic_dctq(tic_ctx *ctx, const uint8_t *image, int h, int w, ptrdiff_t row_stride, int quality,
             int16_t *coeffs_zz);

/* Same stage, reference output convention: dc int32[N] with DPCM applied (codec.py:34-35), ac int32[N*63]. */
int tic_encode(tic_ctx *ctx, const uint8_t *image, int h, int w, ptrdiff_t row_stride, int quality, int32_t *dc,
               int32_t *ac);

/* The same for integer images whose values lie outside 0..255 (the reference casts with astype(int32), codec.py:29, and
 * transforms any integers): int32 pixels (row stride in ELEMENTS), float64 exact operation order on the device, int32
 * coefficients.  A drop-in edge, not a hot path. */
int tic_encode_wide(tic_ctx *ctx, const int32_t *image, int h, int w, ptrdiff_t row_stride_elems, int quality, int32_t *dc,
                    int32_t *ac);
/* encode() / decode() with a non-integral quality (utils.py:50-53 computes with any number; the device holds one constant block
 * per integer quality): installs the constants of `quality`, any number in [1, 99], in the context's spare slot; tic_dctq,
 * tic_encode, tic_encode_wide, tic_dctq_dev and tic_idctq then take quality = TIC_QUALITY_CUSTOM to mean it. */
int tic_set_custom_quality(tic_ctx *ctx, double quality);

/* Device-resident form (what bench.py times): d_image / d_coeffs are device pointers from tic_dev_alloc.
 * Asynchronous on the context's stream; `variant` is one of TIC_KERNEL_*. */
int tic_dev_alloc(tic_ctx *ctx, size_t bytes, void **dptr);
int tic_dev_free(tic_ctx *ctx, void *dptr);
int tic_host_alloc_pinned(tic_ctx *ctx, size_t bytes, void **hptr);
int tic_host_free_pinned(tic_ctx *ctx, void *hptr);
/* Pins memory the caller already holds (hipHostRegister / hipHostUnregister).  tic_compress_batch and tic_dctq_batch copy frames
 * that lie in pinned or registered memory with rows back to back (row_stride == w, w a multiple of 8) to the device from where
 * they are; other frames are staged through the pipeline's own pinned slots (one extra host copy). */
int tic_host_register(tic_ctx *ctx, void *hptr, size_t bytes);
int tic_host_unregister(tic_ctx *ctx, void *hptr);
/* Host side of the batch pipeline and NUMA ("own pinned buffers / NUMA node" per GPU): *node = NUMA node of
 * the context's device (-1: unknown), *ncpus = CPUs of that node the process may run on.  The threads the pipeline creates
 * (staging, read-back, hand-out) bind themselves to those CPUs - never the caller's thread - unless tic_set_numa_binding(ctx, 0);
 * the pinned staging slots are placed on the device's node by hipHostMalloc itself. */
int tic_numa_info(tic_ctx *ctx, int *node, int *ncpus);
int tic_set_numa_binding(tic_ctx *ctx, int enable);
/* How the last batch call took its input: frames copied from the caller's pinned/registered memory / frames staged. */
int tic_last_batch_input_path(tic_ctx *ctx, int *direct_frames, int *staged_frames);
/* Where the last tic_compress_batch / tic_dctq_batch call spent its time: ms8[0] staging copies into pinned memory (pageable
 * input) or registration of the caller's frames, [1] enqueueing (H2D copy, kernels, length copies), [2] waiting for chunks,
 * [3] stream read-back, [4] hand-out into the caller's buffers, [5] waiting for a free slot; [6], [7] unused.  Sums over the
 * call per pipeline thread (0, 1, 5: the submitting thread; 2, 3: the reading thread; 4: the hand-out thread): they overlap. */
int tic_last_batch_phases(tic_ctx *ctx, double *ms8);
/* Pageable frames handed to the batch entry points are pinned IN PLACE for the duration of the call (one hipHostRegister over
 * the address range of the batch, or of a chunk, when the frames lie close together; unregistered before the call returns) and
 * read by the copy engine where they lie; frames that lie scattered, and ranges that cannot be registered, are staged through the
 * pipeline's pinned slots by copy threads.  enable = 0 stages everything (rounds 1-3).  tic_last_batch_auto_registered: how
 * many frames of the last batch call took the registered route (they also count as "direct" in tic_last_batch_input_path). */
int tic_set_auto_register(tic_ctx *ctx, int enable);
int tic_last_batch_auto_registered(tic_ctx *ctx, int *frames);
int tic_memcpy_h2d(tic_ctx *ctx, void *dst, const void *src, size_t bytes);
int tic_memcpy_d2h(tic_ctx *ctx, void *dst, const void *src, size_t bytes);
int tic_memset_dev(tic_ctx *ctx, void *dst, int value, size_t bytes);
int tic_sync(tic_ctx *ctx);
int tic_dctq_dev(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality,
                 void *d_coeffs_zz, int variant);
/* Batch form: `nframes` equally sized frames in ONE launch (grid row per frame).  Frame f starts at
 * d_images + f*frame_stride bytes and its coefficients at d_coeffs_zz + f*coeff_frame_stride bytes. */
int tic_dctq_dev_frames(tic_ctx *ctx, const void *d_images, int nframes, int h, int w, ptrdiff_t row_stride,
                        ptrdiff_t frame_stride, int quality, void *d_coeffs_zz, ptrdiff_t coeff_frame_stride,
                        int variant);
/* Times `iters` back-to-back launches of the transform kernel with HIP events recorded on the context's stream
 * (the stream the kernel is launched on).  *ms_total = elapsed milliseconds for all `iters` launches. */
int tic_dctq_dev_timed(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality,
                       void *d_coeffs_zz, int variant, int iters, float *ms_total);
/* bench.py's form ("hipEvent time of the DCT+quant kernel only, >= 20 iterations after warm-up"): `warm`
 * untimed launches, an event, `iters` timed launches, an event, all in ONE submission - the first event is stamped when the last
 * warm-up launch retires with the timed launches already queued behind it (recorded on an idle stream it is stamped at once and the
 * interval opens with the first launch's submission latency).  per_launch_ms: NULL, or room for 2 * iters floats - then every timed
 * launch carries a start and a stop event on its own dispatch packet (no marker packets between the launches) and
 * per_launch_ms[2i] = duration of launch i, per_launch_ms[2i+1] = start of the first timed launch .. end of launch i. */
int tic_dctq_dev_timed_warm(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality,
                            void *d_coeffs_zz, int variant, int warm, int iters, float *ms_total, float *per_launch_ms);
/* The same for the batch form (nframes frames per launch). */
int tic_dctq_dev_frames_timed(tic_ctx *ctx, const void *d_images, int nframes, int h, int w, ptrdiff_t row_stride,
                              ptrdiff_t frame_stride, int quality, void *d_coeffs_zz, ptrdiff_t coeff_frame_stride,
                              int variant, int iters, float *ms_total);
/* Cold-cache timing: launch i works on pair i % npairs of (d_images[k], d_coeffs_zz[k]); with pairs totalling well over
 * the 256 MiB Infinity Cache every launch streams from and to HBM (bench.py's roofline.cold). */
int tic_dctq_dev_timed_rotating(tic_ctx *ctx, const void *const *d_images, void *const *d_coeffs_zz, int npairs, int h,
                                int w, ptrdiff_t row_stride, int quality, int variant, int iters, float *ms_total);
/* Diagnostics: with stats enabled the HYBRID kernel counts (with a global atomic, which costs time - keep it off
 * when measuring) the blocks it had to redo on the exact path; tic_last_fallback_blocks returns the count
 * accumulated since the previous call and resets it. */
int tic_set_stats(tic_ctx *ctx, int enable);
/* Device entropy stage: which packing kernel.  max_quality < 1 (default): 8 lanes per block for every frame.  max_quality >= 1: a
 * lane per block (a wave = 64 blocks) for qualities up to it; a frame in which a block needs more than 512 bits is transparently
 * packed again with the 8-lane kernel (and the limit then drops below that quality).  Same bytes either way. */
int tic_set_entropy_lane_kernel(tic_ctx *ctx, int max_quality);
int tic_last_fallback_blocks(tic_ctx *ctx, unsigned long long *count);
/* ... and the strip kernel's other rare paths over the same launches (reset by either call): stats[0] = blocks settled in float64 by
 * the batch pass (= tic_last_fallback_blocks), [1] = strips whose rational coefficients were recomputed in the loop (a .5 tie of
 * (0,0) (0,4) (4,0) (4,4) somewhere in the strip), [2] = strips redone as a whole in the exact operation order (batch full, or all
 * eight blocks tripped), [3] = the largest batch any wave carried.  Diagnostics: no counterpart in the reference. */
int tic_last_rare_path_stats(tic_ctx *ctx, unsigned long long stats[4]);

/* ---- entropy stage (host): replaces the per-block loops of compress() codec.py:133-164:
 *      DC DPCM codec.py:34-35, encode_run_length huffman.py:12-33, encode_huffman huffman.py:41-63,
 *      BitBuffer bitbuffer.py:5-72, make_header codec.py:102-114. -------------------------------------------- */
int tic_entropy_encode(const int16_t *coeffs_zz, int h, int w, int quality, uint8_t *out, size_t cap,
                       size_t *out_len);

/* Entropy stage on the device (same stream bytes as tic_entropy_encode): coefficients in HBM -> stream in HBM.
 * Bits per block, a 64-bit offset scan and parallel bit packing; synchronous; cap >= tic_compress_bound(). */
int tic_entropy_encode_dev(tic_ctx *ctx, const void *d_coeffs_zz, int h, int w, int quality, void *d_out, size_t cap,
                           size_t *out_len);

/* ---- whole codec ------------------------------------------------------------------------------------- */
/* compress() with image and stream both resident in HBM: transform kernels + device entropy stage. */
int tic_compress_dev(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality, void *d_out,
                     size_t cap, size_t *out_len);
/* The same, asynchronously: the frame's launches are queued and the call returns with a ticket; up to 64 tickets may be open per
 * context.  For callers that compress resident frames back to back (compress() in a loop, /root/reference/tests/benchmark.py:12-23):
 * submission and completion are paid once per burst, not once per frame, and the transform and packing of a frame run beside the
 * packing and placing of the frame before it (two lanes inside the context).  The streams are WRITTEN in ticket order (the placing
 * kernels run on the context's stream, so anything queued on the context afterwards sees them); until a ticket is collected its input
 * must not be modified and its output not read.  tic_async_result delivers the frame's length or its error (same codes as
 * tic_compress_dev) and closes the ticket; with wait == 0 it returns TIC_E_BUSY while the frame is still in flight (tic_sync waits for
 * everything queued). */
int tic_compress_dev_async(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality, void *d_out,
                           size_t cap, long long *ticket);
int tic_async_result(tic_ctx *ctx, long long ticket, int wait, size_t *out_len);
/* compress() codec.py:133 with auto_generate_huffman_table=False, host buffers: upload, GPU transform stage, GPU
 * entropy stage, download of the finished stream. */
int tic_compress(tic_ctx *ctx, const uint8_t *image, int h, int w, ptrdiff_t row_stride, int quality, uint8_t *out,
                 size_t cap, size_t *out_len);

/* Batch of n independent frames of identical geometry (BASELINE config 3): pinned staging buffers, two HIP
 * streams (the H2D copy of chunk c+1 overlaps the kernels of chunk c and the read-back of chunk c-1).
 * threads <= 0: entropy stage on the device (only finished streams cross PCIe); threads > 0: coefficients are
 * read back and entropy-coded on that many host worker threads.  images[i] / outs[i] are host pointers;
 * out_lens[i] receives each size. */
int tic_compress_batch(tic_ctx *ctx, const uint8_t *const *images, int n, int h, int w, ptrdiff_t row_stride,
                       int quality, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int threads);
/* A batch of one chunk (up to 64 small frames) whose outs[] are the rows of one block of memory (equal distances, 8-byte aligned: a pool of
 * n x cap bytes) has its streams stored straight into that block by the read-back kernel (the block is pinned for the call; off with
 * tic_set_auto_register(ctx, 0)); bytes of outs[i] behind out_lens[i], up to the chunk's longest stream, are then NOT preserved.
 * *streams = how many streams of the last tic_compress_batch went that way. */
int tic_last_batch_zero_copy(tic_ctx *ctx, int *streams);

/* The same batch spread over nctx contexts - one per GPU of the node - by ONE process: a host thread per context, contiguous
 * shards of ceil(n / nctx) frames in frame order, no exchange between the shards; out_lens in frame order.  What a caller that
 * loops over images (/root/reference/tests/benchmark.py:12-23) gets from a multi-GPU node without a launcher.  Returns the first
 * failing shard's code; *failed_ctx (may be null) = that context's index (-1: none).  Contexts must be distinct; two may share a
 * device. */
int tic_compress_batch_multi(tic_ctx *const *ctxs, int nctx, const uint8_t *const *images, int n, int h, int w,
                             ptrdiff_t row_stride, int quality, uint8_t *const *outs, const size_t *caps, size_t *out_lens,
                             int threads, int *failed_ctx);

/* Host threads the batch pipeline may use to stage pageable frames into its pinned slots (default 0: min(8, cores / 2)).  A node
 * that runs one process per GPU sets cores_of_node / local_world_size / 2 here, so that 8 ranks do not start 64 copy threads. */
int tic_set_stage_threads(tic_ctx *ctx, int threads);
int tic_get_stage_threads(tic_ctx *ctx);
/* PCI bus id of the context's device ("0000:c1:00.0"); never NULL. */
const char *tic_pci_bus_id(const tic_ctx *ctx);

/* Same batch, transform stage only (what the metric counts): per-frame coefficients land in coeffs[i]
 * (int16[N*64], host).  If coeffs == NULL the coefficients stay on the device and are discarded. */
int tic_dctq_batch(tic_ctx *ctx, const uint8_t *const *images, int n, int h, int w, ptrdiff_t row_stride,
                   int quality, int16_t *const *coeffs);

/* parse_header() codec.py:117-130 */
int tic_parse_header(const uint8_t *data, size_t len, int *h, int *w, int *quality, uint32_t *flag);

/* decode() codec.py:46-70 from coefficients: coeffs_zz int16[N*64] zig-zag with the DC already integrated
 * (np.cumsum, codec.py:53); GPU dequantise (utils.py:52) + inverse DCT in scipy's operation order
 * (utils.py:40-45) + clip + truncating uint8 cast + crop.  out: uint8[h*w], cap >= h*w. */
int tic_idctq(tic_ctx *ctx, const int16_t *coeffs_zz, int h, int w, int quality, uint8_t *out, size_t cap);

/* decode()'s scaled_dct branch, codec.py:59-62 (coefficients of the reference's C encoder, header flag 1<<30, codec.py:127-128):
 * coeffs / ANNSCALES (constants.py:37-51) * 2**exponent, inverse quantiser of quality 50, then as tic_idctq.  exponent = the
 * stream's quality field, 0..62. */
int tic_idctq_scaled(tic_ctx *ctx, const int16_t *coeffs_zz, int h, int w, int exponent, uint8_t *out, size_t cap);

/* decompress() codec.py:167-189 + decode() codec.py:46-70 for default-table streams, including those of the reference's C
 * encoder (header flag 1<<30 -> scaled_dct branch): host Huffman/RLE decode (huffman.py:36-38,66-98), GPU dequantise +
 * inverse DCT (utils.py:40-45,52) + clip + truncating uint8 cast.  out: uint8[h*w]. */
int tic_decompress(tic_ctx *ctx, const uint8_t *data, size_t len, uint8_t *out, size_t cap);
/* decompress() codec.py:167-189 of `n` streams in one call - the mirror of tic_compress_batch; what the loop of the reference's own
 * benchmark does one stream at a time (tests/benchmark.py:12-23).  Frame i: exactly what tic_decompress(ctx, streams[i], lens[i], outs[i],
 * caps[i]) gives; hs[i] / ws[i] (either array may be null) receive its geometry.  The streams of a chunk go up in one copy, are decoded by
 * ONE launch of each of the device decoder's two kernels, and their pixels come down in one copy (straight into outs[] where the frames
 * follow each other in memory).  Frames the batch kernels do not take - short or damaged streams, C-encoder streams - are decoded by
 * tic_decompress itself behind the batch.  Headers are checked before any work (first bad frame's error, nothing decoded); a decode
 * error is the first failing frame's, all other frames are complete. */
int tic_decompress_batch(tic_ctx *ctx, const uint8_t *const *streams, const size_t *lens, int n, uint8_t *const *outs, const size_t *caps,
                         int *hs, int *ws);
/* How the last tic_decompress_batch went (any pointer may be null): frames decoded by the batch kernels, frames that took the
 * single-frame call, chunks, frames whose pixels were copied straight into the caller's memory. */
int tic_last_decompress_batch(tic_ctx *ctx, int *batch_frames, int *single_frames, int *chunks, int *direct_frames);
/* decompress() with stream and pixels both resident in device memory - the counterpart of tic_compress_dev (decompress()
 * codec.py:167-189 between two device buffers).  d_out receives h rows of w pixels, out_stride bytes apart (out_cap: bytes of
 * the buffer); *h / *w (may be null) receive the geometry of the header.  Long streams never leave the device; short ones and
 * anything the device decoder hands to the host decoder (tic_last_decode_path / _giveup) make the round trip through host memory. */
int tic_decompress_dev(tic_ctx *ctx, const void *d_stream, size_t len, void *d_out, ptrdiff_t out_stride, size_t out_cap, int *h,
                       int *w);
/* Which Huffman decoder the context's last tic_decompress used: 1 = the device decoder (streams of >= 16,384 blocks: only the
 * stream crosses PCIe upwards, only the pixels downwards), 2 = the host decoder (short streams; any long stream in which the
 * device decoder met something unusual - the host's bit-serial path reproduces the reference's behaviour on malformed streams). */
int tic_last_decode_path(tic_ctx *ctx);
/* ... and, when the device decoder handed a long stream to the host decoder, why (0: it did not; bit 1 an incident in the stream's
 * first range, 4 a range without a synchronisation point, 8 / 16 / 32 an incident on the true chain, 2 trace overflow, 64 no block
 * produced). */
int tic_last_decode_giveup(tic_ctx *ctx);
/* ... and the stream bits per lane the device decoder's last run worked with (2 average blocks, 288 ... 2,016; 1,056 at least for nearly flat streams), and how many runs the
 * last long stream took: 2 = the first choice met a range without a synchronisation point and the longest range was tried.
 * Either pointer may be null.  (No counterpart in the reference: huffman.py:77-98 decodes bit by bit.) */
int tic_last_decode_range(tic_ctx *ctx, int *range_bits, int *tries);
/* tic_decompress_dev launches a long stream on a GUESS of its 16-byte header (the header of the stream this context decoded last: the
 * frames of a sequence, the images of a batch; only after two equal headers in a row) instead of reading it from device memory first;
 * the kernels echo the real header and a wrong guess costs a second decode - and nothing else: a run on a wrong guess writes no pixel
 * (the kernel that stores pixels compares the stream's first 16 bytes with the header its geometry came from), so no byte of d_out outside the real h x w is ever touched.  Returns 1 when the last tic_decompress_dev's guess held, -1 when it did not (the stream was
 * decoded again with its own header), 0 when no guess was made.  (No counterpart in the reference: decompress() codec.py:133-164 reads
 * the header from host memory.) */
int tic_last_decode_guess(tic_ctx *ctx);
/* A guess is made only after two streams in a row came with the same header, so alternating geometries never pay for one; enable = 0
 * turns the guessing off altogether for this context (every tic_decompress_dev reads the header first, tic_decompress_dev_async runs
 * synchronously), 1 (the default) back on. */
int tic_set_decode_guess(tic_ctx *ctx, int enable);
/* tic_decompress_dev, asynchronously (the counterpart of tic_compress_dev_async for decompress() in a loop,
 * /root/reference/tests/benchmark.py:19): a long stream is launched on the guess of its header on a stream of the ticket's own and the
 * call returns; up to 4 tickets may be open per context and their frames overlap on the device.  tic_decompress_async_result waits
 * (wait == 0: TIC_E_BUSY while in flight), checks what the kernels reported and, when the guess did not hold or the stream is anything
 * but a whole well-formed one, decodes it again synchronously: a ticket's outcome is always tic_decompress_dev's for the same arguments.
 * The destination must not be read before the result is collected; d_stream must stay valid until then.  Launches are ordered behind
 * everything queued on the context's stream at the time of the call; tic_sync waits for them too. */
int tic_decompress_dev_async(tic_ctx *ctx, const void *d_stream, size_t len, void *d_out, ptrdiff_t out_stride, size_t out_cap,
                             long long *ticket);
int tic_decompress_async_result(tic_ctx *ctx, long long ticket, int wait, int *h_out, int *w_out);

/* ---- multi-GPU (the reference has no counterpart: it is single-process, codec.py:133-164 runs one image
 *      at a time).  One process per GPU; a batch shards by independent frames (frame i -> rank i / ceil(B/G)) with no
 *      data-path collective.  The one exchange is an all-gather of per-frame compressed sizes, so that every rank knows every
 *      frame's offset in the concatenated output: RCCL over xGMI, opened lazily (world == 1 never loads librccl).
 *      Rendezvous: rank 0 publishes the RCCL unique id as the file `rendezvous_path`, the others poll for it; pass a name
 *      unique to the launch and to the communicator (e.g. /tmp/tic_rdv_<MASTER_PORT>_<launcher pid>_<launcher start>_<seq>).
 *      RCCL with more than one rank needs one GPU per rank: it has run with a single rank only so far (no multi-GPU box
 *      was available to the build); the multi-rank flow is covered by world_size-2 tests over gloo. ---------------------- */
typedef struct tic_comm tic_comm;
int tic_comm_create(tic_ctx *ctx, int rank, int world, const char *rendezvous_path, tic_comm **out);
/* The same with the reader's acceptance window spelled out: not_before_ns = the oldest publication time (CLOCK_REALTIME) a
 * reader believes - pass the LAUNCHER's start time when ranks may start long after rank 0 published (staggered launchers, a
 * restarted worker); 0 = the calling process's own start (tic_comm_create), 1 = any age (names inside a directory that is
 * private to the launch).  timeout_ms <= 0 = two minutes. */
int tic_comm_create_ex(tic_ctx *ctx, int rank, int world, const char *rendezvous_path, uint64_t not_before_ns, int timeout_ms,
                       tic_comm **out);
int tic_comm_destroy(tic_comm *comm);
int tic_comm_rank(const tic_comm *comm);
/* NCCL_VERSION_CODE of the RCCL library the communicator runs on (0: single rank, no library loaded).  tic_comm_create refuses a
 * library whose major version is not 2: the RCCL entry points are bound through hand-written RCCL 2.x prototypes. */
int tic_comm_rccl_version(const tic_comm *comm);
int tic_comm_world(const tic_comm *comm);
const char *tic_comm_last_error(const tic_comm *comm);
/* all[r * n_mine + k] = size k of rank r (every rank passes the same n_mine; pad short shards). */
int tic_gather_sizes(tic_comm *comm, const uint64_t *mine, int n_mine, uint64_t *all);
/* In-place element-wise maximum over the ranks (also serves as a barrier: bench.py's max-over-ranks timing). */
int tic_comm_allreduce_max(tic_comm *comm, double *vals, int n);
/* The rendezvous file of tic_comm_create, usable on its own (no GPU involved): publish = exclusive creation (O_EXCL|O_NOFOLLOW,
 * 0600) under a temporary name + rename, after removing leftovers of the same name; wait = poll (timeout_ms) for a complete
 * file OWNED BY THE CALLER'S EFFECTIVE USER with `bytes` of payload published no earlier than not_before_ns (CLOCK_REALTIME;
 * 0 = the calling process's start: a file an earlier launch left behind is ignored; 1 = any age).  The poll interval doubles
 * from 50 us to 20 ms.  tinyimgcodec_amd/distributed.py builds its torch-free control channel (FileComm) on these two. */
int tic_rdv_publish(const char *path, const void *payload, size_t bytes);
int tic_rdv_wait(const char *path, void *payload, size_t bytes, int timeout_ms, uint64_t not_before_ns);

/* ---- diagnostics (not part of the drop-in surface) ------------------------------------------------------------- */
/* Runs the in-register 8x8 byte transpose used by the kernels (DPP + v_perm) and a shuffle-based formulation of
 * the same permutation on `nthreads` x 8 bytes (nthreads multiple of 256); the two outputs must be identical. */
int tic_selftest_transpose(tic_ctx *ctx, const void *host_in, void *host_dpp, void *host_ref, int nthreads);

#ifdef __cplusplus
}
#endif
#endif /* TINYIMGCODEC_HIP_H */
