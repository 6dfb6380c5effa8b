// tic_api.hip - the C-ABI of libtinyimgcodec_hip.so (include/tinyimgcodec_hip.h): context, device buffers,
// launches of the transform kernels on the context's own HIP stream, the stream-overlapped batch pipeline, and
// the glue to the host entropy stage.  No CPU fallback exists for the transform stage.
#include <hip/hip_runtime.h>
#include <chrono>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tinyimgcodec_hip.h"
#include "tic_entropy.h"
#include "tic_entropy_dec_gpu.h"
#include "tic_entropy_gpu.h"
#include "tic_hooks.h"
#include "tic_kernels.h"
#include "tic_math.h"

using namespace tic;

namespace {
constexpr int kAsyncSlots = 64; // tickets of the asynchronous calls that may be open at once per context
constexpr int kDecSlots = 4;    // ... of the asynchronous decodes (each holds a workspace: tens of MB for a 4096^2 stream)
// Phase times of the batch pipeline (tic_last_batch_phases): a handful of steady_clock reads per chunk, summed per context.
// 0 staging copies into pinned memory (pageable input) or registration of the caller's frames, 1 enqueueing (H2D, kernels, lengths),
// 2 waiting for a chunk, 3 stream read-back, 4 hand-out into the caller's buffers, 5 waiting for a free slot.
// (phases 0, 1, 5 belong to the submitting thread, 2 and 3 to the reading thread, 4 to the hand-out thread)
struct BatchTrace {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    static std::chrono::steady_clock::time_point &t0() {
        static thread_local std::chrono::steady_clock::time_point v;
        return v;
    }
    void start() { t0() = std::chrono::steady_clock::now(); }
    void stop(int k) { t[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0()).count(); }
};
#define BT_START() ctx->bt.start()
#define BT_STOP(K) ctx->bt.stop(K)

constexpr size_t kDecTailEnd = 512, kDecTailCoef = 64 * 1024; // device decoder: bytes of stream end prefetched, bytes of tail coefficients (pinned)
constexpr int kChunk = 16;
// ... of small frames more: a chunk of sixteen 512 x 512 frames is 4 MB - its copies and launches cost as much as its work (the reference's
// benchmark set, 49 such frames, took four chunks: 20,000 frames/s); up to 64 frames while a chunk's pixels stay within 32 MB
static inline int chunk_frames(int n, size_t img_bytes) {
    size_t c = img_bytes ? (32u << 20) / img_bytes : (size_t)kChunk;
    c = c < (size_t)kChunk ? (size_t)kChunk : (c > 64 ? 64 : c);
    if (const char *e = test_hook("TIC_BATCH_CHUNK")) c = atoi(e) >= 1 && atoi(e) <= 64 ? (size_t)atoi(e) : c; // (tests: several chunks of small frames)
    return n < (int)c ? n : (int)c;
}
struct Slot {
    uint8_t *pin_in = nullptr;
    int16_t *pin_out = nullptr;
    void *d_img = nullptr;
    void *d_coef = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t rb_done = nullptr; // the chunk's streams have arrived in pin_out (recorded behind the read-back copy)
    // device entropy stage of the chunk
    void *d_work = nullptr;  // workspace of the fused entropy pass
    size_t work_bytes = 0;
    int parity = 0;          // which of the two descriptor arrays / error flags the next call uses
    unsigned long long *d_lens = nullptr, *h_lens = nullptr; // stream length per frame (device / pinned host)
    int *d_err = nullptr, *h_err = nullptr;
    void *d_streams = nullptr;                                // finished streams, one compress_bound() apart
    int first = 0, count = 0; // frames [first, first+count) are in flight in this slot
    int remaining = 0;        // frames of the chunk not yet consumed; 0 = slot free
    size_t rb_row = 0;        // row pitch of the streams read back into pin_out (0: they went straight to the caller)
};
} // namespace

struct tic_ctx {
    // Every entry point that takes a context holds this lock for its whole duration: a context shared between host threads is
    // safe (calls serialise); threads that want to overlap use one context each (the Python mirror's default context is
    // per thread).  Recursive because the host-buffer entry points call the device-buffer ones.
    std::recursive_mutex mu;
    int device = -1;
    hipStream_t stream = nullptr;     // all single-frame work
    hipStream_t bstream[2] = {nullptr, nullptr}; // batch pipeline streams
    hipStream_t rstream = nullptr;               // read-back of finished streams: never queued behind a later chunk's work
    int16_t *h_zz = nullptr;                     // pinned landing buffer of the host Huffman decoder (tic_decompress)
    size_t h_zz_bytes = 0;
    DctqConsts *d_consts = nullptr;   // [100], index = quality; slot 0 = the custom (non-integral) quality of tic_set_custom_quality
    double custom_quality = 0.0;      // what slot 0 holds (0: nothing yet)
    unsigned long long *d_fallback = nullptr;
    bool stats = false; // count guard-band fallbacks with a global atomic (diagnostic; serialises at ~12 ns per wave)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> ev_steps; // tic_dctq_dev_timed_warm with per-launch times: one event behind every timed launch
    // scratch for the host-buffer entry points
    void *d_img = nullptr;
    size_t d_img_cap = 0;
    void *d_coef = nullptr;
    size_t d_coef_cap = 0;
    std::vector<int16_t> h_coef;
    // device entropy stage workspace
    HuffDev *d_huff = nullptr;
    int *d_err = nullptr;
    unsigned long long *d_total_bits = nullptr; // status block of the device entropy stage: payload bits [2], error flags [2] (used in turn)
    unsigned long long *h_stat = nullptr, *d_stat = nullptr; // host-mapped status block of the device entropy stage (bits, error); behind
                                                            // its first 64 bytes: the {bits, error} pairs of the asynchronous calls' tickets
    // asynchronous device-resident calls (tic_compress_dev_async): ticket t lives in slot t % kAsyncSlots
    struct AsyncSlot { long long ticket = -1; size_t cap = 0; int early_rc = TIC_OK; hipEvent_t done = nullptr; bool empty_image = false; };
    AsyncSlot async_slots[kAsyncSlots];
    long long async_next = 0;
    // ... their transform and packing kernels run on two LANES in turn (streams of their own, a coefficient buffer, an entropy workspace and
    // a pair of error flags each), the placing kernels - the only writers of the callers' buffers - on the context's stream in ticket order:
    // frame t + 1 is transformed and packed beside the packing and placing of frame t (tic_compress_dev_async)
    struct AsyncLane {
        hipStream_t stream = nullptr;
        void *d_coef = nullptr;
        size_t coef_cap = 0;
        void *d_work[2] = {nullptr, nullptr}; // two entropy workspaces used in turn: the lane packs its next frame while the placing kernel
        size_t work_bytes = 0;                // of the frame before still reads the other one
        int *d_err = nullptr; // [4], used in turn (a placing kernel zeroes the flag of the lane's frame three ahead: no packing in flight uses it)
        unsigned long long frames = 0;        // frames this lane has taken
        hipEvent_t packed = nullptr, placed[2] = {nullptr, nullptr}; // the lane's last packing has run / the frame that used workspace k has been placed (on ctx->stream)
        bool placed_valid[2] = {false, false};
        unsigned long long order_seen = 0; // the burst (lane_epoch) whose starting point this lane's stream already waits for
    };
    AsyncLane lanes[2];
    hipEvent_t lane_order = nullptr; // a burst's lanes start behind what the context's stream held when the burst began
    int async_open = 0;              // tickets open
    unsigned long long lane_epoch = 0;
    void *d_ent_work = nullptr;                 // workspace of the device entropy stage (tile sums, bit counts, staging slots)
    size_t ent_work_bytes = 0;
    int ent_parity = 0;
    // Which packing kernel the device entropy stage starts with: the lane-per-block kernel up to this quality, the 8-lane kernel
    // above it.  -1 (the default): always the 8-lane kernel - round 3 built the lane-per-block kernel and measured it no faster
    // (4096^2 noise: pack 26 us either way, place 13.5 against 8.5 us; Lenna tiled: 21 + 7 against 24 + 8.5 us; DESIGN.md
    // section 5.4), so it stays selectable (tic_set_entropy_lane_kernel) and tested, not default.  With it on, a frame in which a block needs more
    // than 512 bits (noise at quality >= ~85) makes it raise error 4: the stage is run again with the 8-lane kernel and the limit
    // drops below that quality for the rest of the context's life.
    int ent_lane_max_quality = -1;
    void *d_stream_buf = nullptr;
    size_t d_stream_cap = 0;
    // device Huffman decoder (tic_decompress of long streams): tables, workspace, status
    DecLutsDev *d_dec_luts = nullptr;
    void *d_dec_work = nullptr;
    size_t dec_work_bytes = 0;
    uint8_t *h_dec_tail = nullptr;                              // pinned: kDecTailEnd bytes of stream end + kDecTailCoef bytes of tail coefficients
    unsigned long long *d_dec_desc = nullptr;                   // look-back words of the device decoder's single-launch scans (nothing else lives there)
    size_t dec_desc_words = 0;
    uint32_t dec_epoch = 0;                                     // calls of the device decoder on this workspace (its single-launch scans tell their words by it)
    DecStatus *h_dec_status = nullptr, *d_dec_status = nullptr; // host-mapped
    int last_decode_path = 0;                                  // 0 none, 1 device decoder, 2 host decoder (tic_last_decode_path)
    // asynchronous device-resident decodes (tic_decompress_dev_async): ticket t lives in slot t % kDecSlots; every slot has its own HIP
    // stream, workspace, look-back words and status words, so that the frames of a burst overlap (the measure kernel is one wave per
    // SIMD waiting for table entries, the fused kernel issues float64 arithmetic: they share a CU well)
    struct DecSlot {
        long long ticket = -1;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        void *work = nullptr;
        size_t work_bytes = 0;
        unsigned long long *desc = nullptr;
        size_t desc_words = 0;
        uint32_t epoch = 0;
        DecStatus *h_status = nullptr, *d_status = nullptr;
        bool launched = false;   // false: the call ran synchronously (no guess to launch on): rc / h / w are its outcome
        int rc = TIC_OK, h = 0, w = 0;
        uint8_t head[16] = {0};  // the header the launch guessed
        size_t n = 0;
        const void *d_stream = nullptr; // the call, for the synchronous second try
        size_t len = 0;
        void *d_out = nullptr;
        ptrdiff_t out_stride = 0;
        size_t out_cap = 0;
    };
    DecSlot dec_slots[kDecSlots];
    long long dec_async_next = 0;
    hipEvent_t dec_order = nullptr; // a slot's stream starts behind everything queued on the context's stream so far
    // Small frames to HOST memory (tic_compress): the placing kernel writes the finished stream straight into this host-mapped pinned buffer
    // and the host copies it out with memcpy - no device-to-host DMA copy, whose submission and completion cost more than the transfer of a
    // 30 - 80 KB stream: compress() of a 512 x 512 image 54 us instead of 62 (profiles/r05_decoder.txt).  (The same for the PIXELS of a small
    // decompress(): measured, no gain - 262 KB written across PCIe by the kernel and copied again by the host cost what the DMA copy does.)
    uint8_t *h_small = nullptr, *d_small = nullptr;
    size_t small_cap = 0;
    uint8_t dec_head[16] = {0};                                  // header of the last stream tic_decompress_dev decoded on the device: the next call's guess
    bool dec_head_valid = false;
    int dec_head_streak = 0;   // device decodes in a row (before the last one) whose header was dec_head: a guess is made from 1 on, i.e. after two equal headers
    bool dec_guess_on = true;  // tic_set_decode_guess
    int last_decode_guess = 0;                                  // tic_decompress_dev: 1 the last call's guess of the header held, -1 it did not (decoded again), 0 no guess
    int last_decode_range = 0, last_decode_tries = 0;          // stream bits per lane of the device decoder's last run, and how many runs the last long stream took
    int last_decode_giveup = 0;                                // why the device decoder handed the last long stream to the host (DecStatus::giveup bits)
    // batched decode (tic_decompress_batch): one pinned + one device buffer for a chunk's descriptors and streams, a device and a pinned
    // buffer for its pixels, workspace, look-back words and per-frame status words - kept across calls
    struct DecBatch {
        uint8_t *h_in = nullptr, *d_in = nullptr;
        size_t in_cap = 0;
        uint8_t *d_pix = nullptr, *h_pix = nullptr;
        size_t pix_cap = 0, hpix_cap = 0;
        void *d_work = nullptr;
        size_t work_bytes = 0;
        unsigned long long *d_desc = nullptr;
        size_t desc_words = 0;
        uint32_t epoch = 0;
        DecStatus *h_status = nullptr, *d_status = nullptr;
        size_t status_cap = 0;
    } dbat;
    int last_dbatch_frames = 0, last_dbatch_fallback = 0, last_dbatch_chunks = 0, last_dbatch_direct = 0;
    // batch pipeline buffers, kept across calls (pinned allocations are expensive)
    std::vector<Slot> bslots;
    size_t bslot_img_bytes = 0, bslot_coef_bytes = 0;
    int bslot_h = -1, bslot_w = -1, bslot_chunk = 0;
    // host side of the batch pipeline: the device's NUMA node and the CPUs of that node this process may run on
    int numa_node = -1;
    bool numa_bind = true;      // the pipeline's own threads (staging, read-back, hand-out) bind themselves to those CPUs
    cpu_set_t numa_cpus;
    int numa_ncpus = 0;
    // how the last batch call took its input: frames copied to the device from where the caller holds them (pinned or registered
    // memory) / frames staged through the pipeline's pinned slots (pageable memory)
    int last_batch_direct_frames = 0, last_batch_staged_frames = 0;
    int last_batch_zero_copy = 0; // streams of the last batch the read-back kernel stored straight into the caller's buffers
    mutable BatchTrace bt; // phase times of the last batch call
    // Pageable frames of a batch call are pinned in place for the duration of the call where that is one cheap registration
    // (auto_register_frames) instead of being copied into the pinned slots by CPU threads
    bool auto_register = true;
    std::vector<void *> autoregs;      // ranges this call registered (unregistered before it returns)
    int last_batch_autoreg_frames = 0;
    int stage_threads = 0; // host threads that stage pageable frames (0: min(8, cores / 2)); tic_set_stage_threads
    char pci[32] = {0};
    std::string err;
    char arch[128] = {0};
};

// NUMA node of a device (its PCI function's numa_node in sysfs) and the CPUs of that node within this process's affinity mask.
static void find_numa(tic_ctx *ctx) {
    CPU_ZERO(&ctx->numa_cpus);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, ctx->device) != hipSuccess) return;
    for (char *p = bus; *p; p++) *p = (char)tolower(*p);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return;
    int node = -1;
    const int ok = fscanf(f, "%d", &node);
    fclose(f);
    if (ok != 1 || node < 0) return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return;
    char list[4096] = {0};
    const size_t n = fread(list, 1, sizeof list - 1, f);
    fclose(f);
    list[n] = 0;
    cpu_set_t mine;
    CPU_ZERO(&mine);
    if (sched_getaffinity(0, sizeof mine, &mine) != 0) return;
    // "0-63,128-191": walked with a local cursor (strtol) - contexts are created concurrently, strtok's state is process-global
    for (const char *p = list; *p;) {
        while (*p && (*p < '0' || *p > '9')) p++;
        if (!*p) break;
        char *end = nullptr;
        long a = strtol(p, &end, 10), b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &mine)) { CPU_SET(c, &ctx->numa_cpus); ctx->numa_ncpus++; }
    }
    ctx->numa_node = node;
}
// Called at the start of every thread the pipeline creates (never on the caller's thread).
static void bind_pipeline_thread(const tic_ctx *ctx) {
    // only when the node offers room for the pipeline's threads (8 stagers + reader + hand-out + consumers): a process mask that
    // leaves one or two CPUs of the device's node would put all of them on those
    if (ctx->numa_bind && ctx->numa_node >= 0 && ctx->numa_ncpus >= 8) (void)sched_setaffinity(0, sizeof ctx->numa_cpus, &ctx->numa_cpus);
}

static thread_local std::string g_create_err;

static int set_err(tic_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                                    \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return set_err(ctx, TIC_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                           __LINE__);                                                                        \
    } while (0)

#define TIC_LOCK(ctx)                                      \
    std::unique_lock<std::recursive_mutex> ctx_lock_;      \
    if (ctx) ctx_lock_ = std::unique_lock<std::recursive_mutex>((ctx)->mu)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" {

const char *tic_version(void) { return "tinyimgcodec_amd 0.1.0 (gfx950)"; }
int tic_build_has_test_hooks(void) { return tic::test_hooks_enabled() ? 1 : 0; }

int tic_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *tic_last_error(const tic_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
// (internal, for tic_comm.hip)
__attribute__((visibility("hidden"))) int tic_ctx_device(const tic_ctx *ctx) { return ctx ? ctx->device : -1; }
__attribute__((visibility("hidden"))) void *tic_ctx_stream(const tic_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
const char *tic_device_arch(const tic_ctx *ctx) { return ctx ? ctx->arch : ""; }

size_t tic_num_blocks(int h, int w) { return num_blocks(h, w); }
size_t tic_compress_bound(int h, int w) { return compress_bound(h, w); }

void tic_destroy(tic_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->rstream) (void)hipStreamDestroy(ctx->rstream);
    for (auto &s : ctx->bstream)
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    for (auto &sl : ctx->bslots) {
        if (sl.pin_in) (void)hipHostFree(sl.pin_in);
        if (sl.pin_out) (void)hipHostFree(sl.pin_out);
        if (sl.d_img) (void)hipFree(sl.d_img);
        if (sl.d_coef) (void)hipFree(sl.d_coef);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.d_work) (void)hipFree(sl.d_work);
        if (sl.d_lens) (void)hipFree(sl.d_lens);
        if (sl.h_lens) (void)hipHostFree(sl.h_lens);
        if (sl.d_err) (void)hipFree(sl.d_err);
        if (sl.h_err) (void)hipHostFree(sl.h_err);
        if (sl.d_streams) (void)hipFree(sl.d_streams);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (auto &e : ctx->ev_steps)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_img) (void)hipFree(ctx->d_img);
    if (ctx->d_coef) (void)hipFree(ctx->d_coef);
    if (ctx->d_consts) (void)hipFree(ctx->d_consts);
    if (ctx->d_fallback) (void)hipFree(ctx->d_fallback);
    if (ctx->d_huff) (void)hipFree(ctx->d_huff);
    if (ctx->d_total_bits) (void)hipFree(ctx->d_total_bits); // d_err lives in the same block
    if (ctx->d_ent_work) (void)hipFree(ctx->d_ent_work);
    for (auto &sl : ctx->async_slots)
        if (sl.done) (void)hipEventDestroy(sl.done);
    for (auto &ln : ctx->lanes) {
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
        if (ln.d_coef) (void)hipFree(ln.d_coef);
        for (int k = 0; k < 2; k++) {
            if (ln.d_work[k]) (void)hipFree(ln.d_work[k]);
            if (ln.placed[k]) (void)hipEventDestroy(ln.placed[k]);
        }
        if (ln.d_err) (void)hipFree(ln.d_err);
        if (ln.packed) (void)hipEventDestroy(ln.packed);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    if (ctx->lane_order) (void)hipEventDestroy(ctx->lane_order);
    if (ctx->h_stat) (void)hipHostFree(ctx->h_stat);
    if (ctx->h_zz) (void)hipHostFree(ctx->h_zz);
    if (ctx->d_stream_buf) (void)hipFree(ctx->d_stream_buf);
    if (ctx->d_dec_luts) (void)hipFree(ctx->d_dec_luts);
    if (ctx->d_dec_work) (void)hipFree(ctx->d_dec_work);
    if (ctx->d_dec_desc) (void)hipFree(ctx->d_dec_desc);
    if (ctx->h_dec_tail) (void)hipHostFree(ctx->h_dec_tail);
    if (ctx->h_dec_status) (void)hipHostFree(ctx->h_dec_status);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    for (auto &sl : ctx->dec_slots) {
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.work) (void)hipFree(sl.work);
        if (sl.desc) (void)hipFree(sl.desc);
        if (sl.h_status) (void)hipHostFree(sl.h_status);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (ctx->dbat.h_in) (void)hipHostFree(ctx->dbat.h_in);
    if (ctx->dbat.d_in) (void)hipFree(ctx->dbat.d_in);
    if (ctx->dbat.d_pix) (void)hipFree(ctx->dbat.d_pix);
    if (ctx->dbat.h_pix) (void)hipHostFree(ctx->dbat.h_pix);
    if (ctx->dbat.d_work) (void)hipFree(ctx->dbat.d_work);
    if (ctx->dbat.d_desc) (void)hipFree(ctx->dbat.d_desc);
    if (ctx->dbat.h_status) (void)hipHostFree(ctx->dbat.h_status);
    if (ctx->dec_order) (void)hipEventDestroy(ctx->dec_order);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int create_impl(tic_ctx *ctx, int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_err(nullptr, TIC_E_NODEVICE, "no HIP device available (%s): the transform stage has no CPU fallback",
                       e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return set_err(nullptr, TIC_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    ctx->device = device;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return set_err(nullptr, TIC_E_HIP, "cannot open device %d: %s", device, hipGetErrorString(e));
    snprintf(ctx->arch, sizeof ctx->arch, "%s", prop.gcnArchName);
    if (hipDeviceGetPCIBusId(ctx->pci, sizeof ctx->pci, device) != hipSuccess) {
        (void)hipGetLastError();
        ctx->pci[0] = 0;
    }
    find_numa(ctx);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(nullptr, TIC_E_NODEVICE, "device %d is %s; this library contains gfx950 (MI355X) code only", device,
                       prop.gcnArchName);
#define CK(call)                                                                                          \
    if ((e = (call)) != hipSuccess)                                                                       \
        return set_err(nullptr, TIC_E_HIP, "%s failed: %s", #call, hipGetErrorString(e));
    CK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ctx->bstream[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ctx->bstream[1], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ctx->rstream, hipStreamNonBlocking));
    CK(hipEventCreate(&ctx->ev0));
    CK(hipEventCreate(&ctx->ev1));
    std::vector<DctqConsts> all(100);
    memset(all.data(), 0, all.size() * sizeof(DctqConsts));
    for (int q = 1; q <= 99; q++) build_consts(q, &all[q]);
    CK(hipMalloc((void **)&ctx->d_consts, all.size() * sizeof(DctqConsts)));
    CK(hipMemcpy(ctx->d_consts, all.data(), all.size() * sizeof(DctqConsts), hipMemcpyHostToDevice));
    {
        HuffDev hd;
        build_huff_dev(&hd);
        CK(hipMalloc((void **)&ctx->d_huff, sizeof(HuffDev)));
        CK(hipMemcpy(ctx->d_huff, &hd, sizeof(HuffDev), hipMemcpyHostToDevice));
        // one 32-byte status block: payload bits [2] of the device entropy stage, then its error flags [2]
        CK(hipMalloc((void **)&ctx->d_total_bits, 32));
        CK(hipMemset(ctx->d_total_bits, 0, 32));
        ctx->d_err = reinterpret_cast<int *>(ctx->d_total_bits + 2);
        // the stage's last kernel writes {payload bits, error} straight into pinned host memory: no copy behind it
        CK(hipHostMalloc((void **)&ctx->h_stat, 64 + kAsyncSlots * 16, hipHostMallocMapped | hipHostMallocCoherent));
        memset(ctx->h_stat, 0, 64 + kAsyncSlots * 16);
        CK(hipHostGetDevicePointer((void **)&ctx->d_stat, ctx->h_stat, 0));
    }
    CK(hipMalloc((void **)&ctx->d_fallback, 4 * sizeof(unsigned long long)));
    CK(hipMemset(ctx->d_fallback, 0, 4 * sizeof(unsigned long long)));
#undef CK
    return TIC_OK;
}

tic_ctx *tic_create(int device) {
    tic_ctx *ctx = new tic_ctx();
    if (create_impl(ctx, device) != TIC_OK) {
        if (ctx->device >= 0) tic_destroy(ctx); else delete ctx;
        return nullptr;
    }
    return ctx;
}

// ---- device memory helpers ---------------------------------------------------------------------------
int tic_dev_alloc(tic_ctx *ctx, size_t bytes, void **dptr) {
    TIC_LOCK(ctx);
    if (!ctx || !dptr) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return TIC_OK;
}
int tic_dev_free(tic_ctx *ctx, void *dptr) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipFree(dptr));
    return TIC_OK;
}
int tic_host_alloc_pinned(tic_ctx *ctx, size_t bytes, void **hptr) {
    TIC_LOCK(ctx);
    if (!ctx || !hptr) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return TIC_OK;
}
int tic_host_free_pinned(tic_ctx *ctx, void *hptr) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostFree(hptr));
    return TIC_OK;
}
// Pins memory the caller already holds (hipHostRegister): the batch entry points then copy such frames to the device from where
// they lie instead of staging them through their own pinned slots.
int tic_host_register(tic_ctx *ctx, void *hptr, size_t bytes) {
    TIC_LOCK(ctx);
    if (!ctx || !hptr || bytes == 0) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostRegister(hptr, bytes, hipHostRegisterDefault));
    return TIC_OK;
}
int tic_host_unregister(tic_ctx *ctx, void *hptr) {
    TIC_LOCK(ctx);
    if (!ctx || !hptr) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostUnregister(hptr));
    return TIC_OK;
}
// NUMA placement of the batch pipeline's host side.  *node = NUMA node of the context's device (-1 unknown), *ncpus = CPUs of that
// node this process may run on.  The pipeline's own threads bind to them unless binding is switched off (enable = 0); the pinned
// staging slots come from hipHostMalloc, which places them on the device's node by itself.
int tic_numa_info(tic_ctx *ctx, int *node, int *ncpus) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    if (node) *node = ctx->numa_node;
    if (ncpus) *ncpus = ctx->numa_ncpus;
    return TIC_OK;
}
int tic_set_stage_threads(tic_ctx *ctx, int threads) {
    TIC_LOCK(ctx);
    if (!ctx || threads < 0) return TIC_E_ARG;
    ctx->stage_threads = threads > 8 ? 8 : threads;
    return TIC_OK;
}
int tic_get_stage_threads(tic_ctx *ctx) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    if (ctx->stage_threads > 0) return ctx->stage_threads;
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = (int)(hw ? hw / 2 : 4);
    return T < 1 ? 1 : (T > 8 ? 8 : T);
}
const char *tic_pci_bus_id(const tic_ctx *ctx) { return ctx ? ctx->pci : ""; }

int tic_set_numa_binding(tic_ctx *ctx, int enable) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    ctx->numa_bind = enable != 0;
    return TIC_OK;
}
// How the last tic_compress_batch / tic_dctq_batch call took its frames: copied from the caller's pinned or registered memory /
// staged through the pipeline's slots.
int tic_last_batch_input_path(tic_ctx *ctx, int *direct_frames, int *staged_frames) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    if (direct_frames) *direct_frames = ctx->last_batch_direct_frames;
    if (staged_frames) *staged_frames = ctx->last_batch_staged_frames;
    return TIC_OK;
}
// Pageable frames of a batch are pinned in place for the duration of the call where one registration covers them
// (auto_register_frames); enable = 0 stages them through the pipeline's pinned slots as rounds 1-3 did.
int tic_set_auto_register(tic_ctx *ctx, int enable) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    ctx->auto_register = enable != 0;
    return TIC_OK;
}
int tic_last_batch_auto_registered(tic_ctx *ctx, int *frames) {
    TIC_LOCK(ctx);
    if (!ctx || !frames) return TIC_E_ARG;
    *frames = ctx->last_batch_autoreg_frames;
    return TIC_OK;
}
int tic_last_batch_zero_copy(tic_ctx *ctx, int *streams) {
    TIC_LOCK(ctx);
    if (!ctx || !streams) return TIC_E_ARG;
    *streams = ctx->last_batch_zero_copy;
    return TIC_OK;
}
int tic_last_batch_phases(tic_ctx *ctx, double *ms8) {
    TIC_LOCK(ctx);
    if (!ctx || !ms8) return TIC_E_ARG;
    for (int k = 0; k < 8; k++) ms8[k] = ctx->bt.t[k] * 1e3;
    return TIC_OK;
}

int tic_memcpy_h2d(tic_ctx *ctx, void *dst, const void *src, size_t bytes) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TIC_OK;
}
int tic_memcpy_d2h(tic_ctx *ctx, void *dst, const void *src, size_t bytes) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TIC_OK;
}
int tic_memset_dev(tic_ctx *ctx, void *dst, int value, size_t bytes) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return TIC_OK;
}
int tic_sync(tic_ctx *ctx) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &sl : ctx->dec_slots) // (asynchronous decodes run on streams of their own)
        if (sl.stream) HIPCHK(ctx, hipStreamSynchronize(sl.stream));
    return TIC_OK;
}

// ---- transform stage -----------------------------------------------------------------------------------
// Waits for the context's stream.  A blocking hipStreamSynchronize wakes the thread some microseconds after the last kernel
// retired - a sixth of a 60 us device-resident compress - so short waits poll the stream first and only long ones block.
static hipError_t wait_stream(tic_ctx *ctx) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(250)) break;
    }
    return hipStreamSynchronize(ctx->stream);
}

static int check_geometry(tic_ctx *ctx, int h, int w, ptrdiff_t stride, int quality) {
    if (!ctx) return TIC_E_ARG;
    if (h < 0 || w < 0) return set_err(ctx, TIC_E_ARG, "negative image size %dx%d", h, w);
    if ((quality < 1 || quality > 99) && !(quality == TIC_QUALITY_CUSTOM && ctx->custom_quality != 0.0))
        return set_err(ctx, TIC_E_QUALITY, "quality %d outside 1..99 (the reference fails for 0, 100 and negatives)",
                       quality);
    if (h > 0 && w > 0 && stride < (ptrdiff_t)w) return set_err(ctx, TIC_E_ARG, "row stride %td < width %d", stride, w);
    return TIC_OK;
}

// ... for the entry points that write a stream: its header holds the quality as an integer (codec.py:102-114), the custom slot has none
static int check_stream_geometry(tic_ctx *ctx, int h, int w, ptrdiff_t stride, int quality) {
    if (ctx && quality == TIC_QUALITY_CUSTOM) return set_err(ctx, TIC_E_QUALITY, "a stream needs an integer quality 1..99 in its header");
    return check_geometry(ctx, h, w, stride, quality);
}

static DctqArgs make_args(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t stride, int quality, void *d_out) {
    DctqArgs a;
    a.img = (const uint8_t *)d_image;
    a.h = h;
    a.w = w;
    a.stride = (long)stride;
    a.bw = (w + 7) / 8;
    a.tiles_x = (a.bw + 7) / 8;
    a.ntiles = (h > 0 && w > 0) ? ((h + 7) / 8) * a.tiles_x : 0;
    a.aligned8 = ((((uintptr_t)d_image) | (uintptr_t)stride) & 7) == 0;
    a.consts = ctx->d_consts + quality;
    a.out = (int16_t *)d_out;
    a.fallback_count = ctx->stats ? ctx->d_fallback : nullptr;
    a.nframes = 1;
    a.frame_stride_in = 0;
    a.frame_stride_out = 0;
    a.dbg = nullptr;
    return a;
}

// Frames that follow each other without a gap and end on a block row ARE one tall frame for the transform stage (the DC is not
// differenced there): one persistent grid walks the whole batch in chunks, as for a 16384^2 frame, instead of a grid plane per
// frame with its own ramp and tail (256 x 1080p: 0.63-0.67 of the roofline per plane, see DESIGN.md section 5.5 for the merged form).
static void merge_frames(DctqArgs &a) {
    if (a.nframes <= 1 || (a.h & 7) != 0) return;
    const long long nblk = (long long)(a.h / 8) * a.bw;
    if (a.frame_stride_in != (long)a.h * a.stride || a.frame_stride_out != (long)(nblk * 128)) return;
    if ((unsigned long long)a.frame_stride_in * (unsigned long long)a.nframes >= (1ull << 32)) return; // the strip walk uses 32-bit offsets
    if ((long long)a.h * a.nframes > 0x7fffffffll / 8 || nblk * a.nframes > 0x7fffffffll) return;
    a.h *= a.nframes;
    a.ntiles = ((a.h + 7) / 8) * a.tiles_x;
    a.nframes = 1;
    a.frame_stride_in = 0;
    a.frame_stride_out = 0;
}

int tic_dctq_dev(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality, void *d_coeffs_zz,
                 int variant) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (h == 0 || w == 0) return TIC_OK;
    if (!d_image || !d_coeffs_zz) return set_err(ctx, TIC_E_ARG, "null device pointer");
    const int v = dctq_kernel_id(variant);
    if (v < 0) return set_err(ctx, TIC_E_ARG, "unknown kernel variant %d", variant);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DctqArgs a = make_args(ctx, d_image, h, w, row_stride, quality, d_coeffs_zz);
    HIPCHK(ctx, launch_dctq(a, v, ctx->stream));
    return TIC_OK;
}

int tic_dctq_dev_frames(tic_ctx *ctx, const void *d_images, int nframes, int h, int w, ptrdiff_t row_stride,
                        ptrdiff_t frame_stride, int quality, void *d_coeffs_zz, ptrdiff_t coeff_frame_stride, int variant) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (nframes < 0) return set_err(ctx, TIC_E_ARG, "negative frame count");
    if (h == 0 || w == 0 || nframes == 0) return TIC_OK;
    if (!d_images || !d_coeffs_zz) return set_err(ctx, TIC_E_ARG, "null device pointer");
    if (nframes > 65535) return set_err(ctx, TIC_E_ARG, "at most 65535 frames per launch");
    if (frame_stride < (ptrdiff_t)h * row_stride || coeff_frame_stride < (ptrdiff_t)(num_blocks(h, w) * 128))
        return set_err(ctx, TIC_E_ARG, "frame strides smaller than one frame");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DctqArgs a = make_args(ctx, d_images, h, w, row_stride, quality, d_coeffs_zz);
    a.aligned8 = a.aligned8 && ((frame_stride & 7) == 0);
    a.nframes = nframes;
    a.frame_stride_in = (long)frame_stride;
    a.frame_stride_out = (long)coeff_frame_stride;
    merge_frames(a);
    const int v = dctq_kernel_id(variant);
    if (v < 0) return set_err(ctx, TIC_E_ARG, "unknown kernel variant %d", variant);
    HIPCHK(ctx, launch_dctq(a, v, ctx->stream));
    return TIC_OK;
}

int tic_dctq_dev_timed(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality,
                       void *d_coeffs_zz, int variant, int iters, float *ms_total) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (!ms_total || iters < 1 || !d_image || !d_coeffs_zz) return set_err(ctx, TIC_E_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DctqArgs a = make_args(ctx, d_image, h, w, row_stride, quality, d_coeffs_zz);
    const int v = dctq_kernel_id(variant);
    if (v < 0) return set_err(ctx, TIC_E_ARG, "unknown kernel variant %d", variant);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; i++) HIPCHK(ctx, launch_dctq(a, v, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return TIC_OK;
}

// The benchmark's form of the timed entry: `warm` untimed launches, an event, `iters` timed launches, an event - ONE submission,
// nothing between the warm-up and the first event that the host waits for.  The first event is therefore stamped when the last
// warm-up launch retires, with the timed launches already in the queue behind it; recorded on an IDLE stream (tic_dctq_dev_timed
// after a synchronisation) it is stamped at once and the interval opens with whatever the host needs to get the first launch to
// the device - 28 us on the driver's box in round 5, 1.4 us per step at K = 20 (profiles/r06_driver_flags.txt).
// per_launch_ms: NULL, or room for 2 * iters floats.  Then every timed launch carries a start and a stop event on its OWN dispatch
// packet (hipExtLaunchKernelGGL: the packet's time stamps, no marker packet between the launches - an event recorded behind every
// launch costs 3 us per launch) and per_launch_ms[2 i] = duration of launch i, per_launch_ms[2 i + 1] = the time between the start
// of the first timed launch and the end of launch i.
int tic_dctq_dev_timed_warm(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality, void *d_coeffs_zz,
                            int variant, int warm, int iters, float *ms_total, float *per_launch_ms) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (!ms_total || iters < 1 || warm < 0 || !d_image || !d_coeffs_zz) return set_err(ctx, TIC_E_ARG, "bad argument");
    if (per_launch_ms && iters > 32768) return set_err(ctx, TIC_E_ARG, "per-launch times for at most 32768 launches");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DctqArgs a = make_args(ctx, d_image, h, w, row_stride, quality, d_coeffs_zz);
    const int v = dctq_kernel_id(variant);
    if (v < 0) return set_err(ctx, TIC_E_ARG, "unknown kernel variant %d", variant);
    if (per_launch_ms)
        while ((int)ctx->ev_steps.size() < 2 * iters) {
            hipEvent_t e = nullptr;
            HIPCHK(ctx, hipEventCreate(&e));
            ctx->ev_steps.push_back(e);
        }
    for (int i = 0; i < warm; i++) HIPCHK(ctx, launch_dctq(a, v, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; i++) {
        if (per_launch_ms)
            HIPCHK(ctx, launch_dctq(a, v, ctx->stream, ctx->ev_steps[2 * i], ctx->ev_steps[2 * i + 1]));
        else
            HIPCHK(ctx, launch_dctq(a, v, ctx->stream));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    if (per_launch_ms)
        for (int i = 0; i < iters; i++) {
            // (a frame that does not take the strip kernel - padded strips only, bands - leaves its events unrecorded: reported as such)
            if (hipEventElapsedTime(per_launch_ms + 2 * i, ctx->ev_steps[2 * i], ctx->ev_steps[2 * i + 1]) != hipSuccess ||
                hipEventElapsedTime(per_launch_ms + 2 * i + 1, ctx->ev_steps[0], ctx->ev_steps[2 * i + 1]) != hipSuccess) {
                (void)hipGetLastError();
                return set_err(ctx, TIC_E_ARG, "per-launch times need a frame that takes the strip kernel in one launch");
            }
        }
    return TIC_OK;
}

// Batch form of the timed entry: `iters` back-to-back launches of the batched transform (nframes frames per launch).
int tic_dctq_dev_frames_timed(tic_ctx *ctx, const void *d_images, int nframes, int h, int w, ptrdiff_t row_stride,
                              ptrdiff_t frame_stride, int quality, void *d_coeffs_zz, ptrdiff_t coeff_frame_stride, int variant,
                              int iters, float *ms_total) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (!ms_total || iters < 1 || nframes < 1 || nframes > 65535 || !d_images || !d_coeffs_zz) return set_err(ctx, TIC_E_ARG, "bad argument");
    const int v = dctq_kernel_id(variant);
    if (v < 0) return set_err(ctx, TIC_E_ARG, "unknown kernel variant %d", variant);
    if (frame_stride < (ptrdiff_t)h * row_stride || coeff_frame_stride < (ptrdiff_t)(num_blocks(h, w) * 128))
        return set_err(ctx, TIC_E_ARG, "frame strides smaller than one frame");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DctqArgs a = make_args(ctx, d_images, h, w, row_stride, quality, d_coeffs_zz);
    a.aligned8 = a.aligned8 && ((frame_stride & 7) == 0);
    a.nframes = nframes;
    a.frame_stride_in = (long)frame_stride;
    a.frame_stride_out = (long)coeff_frame_stride;
    merge_frames(a);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; i++) HIPCHK(ctx, launch_dctq(a, v, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return TIC_OK;
}

// Cold-cache form of the timed entry: launch i works on pair i % npairs of (image, coefficient buffer).  With enough
// distinct pairs (their total size well beyond the 256 MiB Infinity Cache) every launch reads and writes lines that have
// left the cache since their last use: the number is an HBM number, not an on-die one.
int tic_dctq_dev_timed_rotating(tic_ctx *ctx, const void *const *d_images, void *const *d_coeffs_zz, int npairs, int h, int w,
                                ptrdiff_t row_stride, int quality, int variant, int iters, float *ms_total) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (!ms_total || iters < 1 || npairs < 1 || !d_images || !d_coeffs_zz) return set_err(ctx, TIC_E_ARG, "bad argument");
    const int v = dctq_kernel_id(variant);
    if (v < 0) return set_err(ctx, TIC_E_ARG, "unknown kernel variant %d", variant);
    for (int k = 0; k < npairs; k++)
        if (!d_images[k] || !d_coeffs_zz[k]) return set_err(ctx, TIC_E_ARG, "null device pointer in pair %d", k);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; i++) {
        DctqArgs a = make_args(ctx, d_images[i % npairs], h, w, row_stride, quality, d_coeffs_zz[i % npairs]);
        HIPCHK(ctx, launch_dctq(a, v, ctx->stream));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return TIC_OK;
}

// Device entropy stage: pack with a lane per block (max_quality >= 1: for qualities up to it, with the automatic fall-back to the
// 8-lane kernel described at ent_lane_max_quality) or always with 8 lanes per block (max_quality < 1, the default).
int tic_set_entropy_lane_kernel(tic_ctx *ctx, int max_quality) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    ctx->ent_lane_max_quality = max_quality < 1 ? -1 : (max_quality > 99 ? 99 : max_quality);
    return TIC_OK;
}

// Non-integral qualities.  The reference's encode() / decode() compute with whatever number they are given (utils.py:50-53:
// factor = 5000 / q or 200 - 2 q, divisor = Q * factor / 100); the device keeps one constant block per INTEGER quality.  This
// installs the block of any number in [1, 99] in the context's spare slot; the transform and inverse entry points (tic_dctq,
// tic_encode, tic_encode_wide, tic_dctq_dev, tic_idctq) then take quality = TIC_QUALITY_CUSTOM to mean it.  (compress() packs the
// quality into the header as an integer - struct.error for a float in the reference - so the stream entry points have no use for it.)
int tic_set_custom_quality(tic_ctx *ctx, double quality) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    std::unique_ptr<DctqConsts> c(new DctqConsts());
    if (!build_consts(quality, c.get())) return set_err(ctx, TIC_E_QUALITY, "quality %g outside 1..99", quality);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // (nothing in flight reads the slot: every entry point that uses it ends drained)
    HIPCHK(ctx, hipMemcpy(ctx->d_consts, c.get(), sizeof(DctqConsts), hipMemcpyHostToDevice));
    ctx->custom_quality = quality;
    return TIC_OK;
}

int tic_set_stats(tic_ctx *ctx, int enable) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    ctx->stats = enable != 0;
    return TIC_OK;
}

int tic_last_rare_path_stats(tic_ctx *ctx, unsigned long long stats[4]) {
    TIC_LOCK(ctx);
    if (!ctx || !stats) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(stats, ctx->d_fallback, 4 * sizeof *stats, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_fallback, 0, 4 * sizeof *stats, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TIC_OK;
}

int tic_last_fallback_blocks(tic_ctx *ctx, unsigned long long *count) {
    if (!ctx || !count) return TIC_E_ARG;
    unsigned long long st[4];
    const int rc = tic_last_rare_path_stats(ctx, st);
    if (rc == TIC_OK) *count = st[0];
    return rc;
}

constexpr size_t kDecHostPixBytes = 1u << 20; // tic_decompress: images of at most this many bytes leave through tic_ctx::h_small as well
constexpr size_t kSmallHostBytes = 2u << 20; // tic_compress: frames whose stream bound is at most this go through tic_ctx::h_small
static int ensure_small(tic_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->small_cap) return TIC_OK;
    if (ctx->h_small) HIPCHK(ctx, hipHostFree(ctx->h_small));
    ctx->h_small = ctx->d_small = nullptr;
    ctx->small_cap = 0;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_small, bytes, hipHostMallocMapped | hipHostMallocCoherent)); // (read by the host behind a polled stream)
    HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->d_small, ctx->h_small, 0));
    ctx->small_cap = bytes;
    return TIC_OK;
}

static int ensure_scratch(tic_ctx *ctx, size_t img_bytes, size_t coef_bytes) {
    if (img_bytes > ctx->d_img_cap) {
        if (ctx->d_img) HIPCHK(ctx, hipFree(ctx->d_img));
        ctx->d_img = nullptr;
        ctx->d_img_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_img, img_bytes));
        ctx->d_img_cap = img_bytes;
    }
    if (coef_bytes > ctx->d_coef_cap) {
        if (ctx->d_coef) HIPCHK(ctx, hipFree(ctx->d_coef));
        ctx->d_coef = nullptr;
        ctx->d_coef_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_coef, coef_bytes));
        ctx->d_coef_cap = coef_bytes;
    }
    return TIC_OK;
}

int tic_dctq(tic_ctx *ctx, const uint8_t *image, int h, int w, ptrdiff_t row_stride, int quality, int16_t *coeffs_zz) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    const size_t n = num_blocks(h, w);
    if (n == 0) return TIC_OK;
    if (!image || !coeffs_zz) return set_err(ctx, TIC_E_ARG, "null host pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t pitch = align_up((size_t)w, 256);
    rc = ensure_scratch(ctx, pitch * (size_t)h, n * 128);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_img, pitch, image, (size_t)row_stride, (size_t)w, (size_t)h,
                                 hipMemcpyHostToDevice, ctx->stream));
    DctqArgs a = make_args(ctx, ctx->d_img, h, w, (ptrdiff_t)pitch, quality, ctx->d_coef);
    HIPCHK(ctx, launch_dctq(a, 2, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(coeffs_zz, ctx->d_coef, n * 128, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TIC_OK;
}

int tic_encode(tic_ctx *ctx, const uint8_t *image, int h, int w, ptrdiff_t row_stride, int quality, int32_t *dc,
               int32_t *ac) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    const size_t n = num_blocks(h, w);
    if (n == 0) return TIC_OK;
    if (!dc || !ac) return set_err(ctx, TIC_E_ARG, "null output pointer");
    ctx->h_coef.resize(n * 64);
    rc = tic_dctq(ctx, image, h, w, row_stride, quality, ctx->h_coef.data());
    if (rc) return rc;
    int prev = 0;
    for (size_t b = 0; b < n; b++) { // codec.py:34-36
        const int16_t *c = ctx->h_coef.data() + b * 64;
        dc[b] = b ? c[0] - prev : c[0];
        prev = c[0];
        for (int k = 1; k < 64; k++) ac[b * 63 + (k - 1)] = c[k];
    }
    return TIC_OK;
}

// encode() for integer images outside 0..255 (codec.py:29 casts with astype(int32) and transforms whatever it finds):
// int32 pixels, float64 exact order on the device, int32 dc (DPCM applied) / ac as the reference returns them.
int tic_encode_wide(tic_ctx *ctx, const int32_t *image, int h, int w, ptrdiff_t row_stride_elems, int quality, int32_t *dc,
                    int32_t *ac) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride_elems, quality);
    if (rc) return rc;
    const size_t n = num_blocks(h, w);
    if (n == 0) return TIC_OK;
    if (!image || !dc || !ac) return set_err(ctx, TIC_E_ARG, "null pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rc = ensure_scratch(ctx, (size_t)h * (size_t)w * 4, n * 256);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_img, (size_t)w * 4, image, (size_t)row_stride_elems * 4, (size_t)w * 4, (size_t)h,
                                 hipMemcpyHostToDevice, ctx->stream));
    WideArgs a;
    a.img = (const int32_t *)ctx->d_img;
    a.out = (int32_t *)ctx->d_coef;
    a.h = h;
    a.w = w;
    a.stride = w;
    a.bw = (w + 7) / 8;
    a.tiles_x = (a.bw + 7) / 8;
    a.ntiles = ((h + 7) / 8) * a.tiles_x;
    a.consts = ctx->d_consts + quality;
    HIPCHK(ctx, launch_dctq_wide(a, ctx->stream));
    std::vector<int32_t> zz(n * 64);
    HIPCHK(ctx, hipMemcpyAsync(zz.data(), ctx->d_coef, n * 256, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int32_t prev = 0;
    for (size_t b = 0; b < n; b++) { // codec.py:34-36 (np.diff wraps in int32 like the reference's arrays)
        const int32_t *c = zz.data() + b * 64;
        dc[b] = b ? (int32_t)((uint32_t)c[0] - (uint32_t)prev) : c[0];
        prev = c[0];
        for (int k = 1; k < 64; k++) ac[b * 63 + (k - 1)] = c[k];
    }
    return TIC_OK;
}

// ---- entropy stage / whole codec -------------------------------------------------------------------------
int tic_entropy_encode(const int16_t *coeffs_zz, int h, int w, int quality, uint8_t *out, size_t cap, size_t *out_len) {
    return entropy_encode(coeffs_zz, h, w, quality, out, cap, out_len);
}

int tic_parse_header(const uint8_t *data, size_t len, int *h, int *w, int *quality, uint32_t *flag) {
    return parse_header(data, len, h, w, quality, flag);
}

// Device entropy stage: coefficients in HBM -> finished stream in HBM.  Synchronous (the stream length is needed
// on the host between the counting and the packing step).
int tic_entropy_encode_dev(tic_ctx *ctx, const void *d_coeffs_zz, int h, int w, int quality, void *d_out, size_t cap,
                           size_t *out_len) {
    TIC_LOCK(ctx);
    if (!ctx || !out_len) return TIC_E_ARG;
    if (h < 0 || w < 0) return set_err(ctx, TIC_E_ARG, "negative image size");
    if (quality < 1 || quality > 99) return set_err(ctx, TIC_E_QUALITY, "quality %d outside 1..99", quality);
    if (!d_out || cap < 16) return set_err(ctx, TIC_E_SPACE, "output buffer too small");
    if (((uintptr_t)d_out & 15u) != 0) return set_err(ctx, TIC_E_ARG, "device output buffer must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = num_blocks(h, w);
    if (n == 0) {
        uint8_t hdr[16];
        write_header(hdr, h, w, quality);
        HIPCHK(ctx, hipMemcpyAsync(d_out, hdr, 16, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *out_len = 16;
        return TIC_OK;
    }
    if (!d_coeffs_zz) return set_err(ctx, TIC_E_ARG, "null coefficient pointer");
    const size_t wb = entropy_fused_work_bytes(n);
    if (wb > ctx->ent_work_bytes) {
        if (ctx->d_ent_work) HIPCHK(ctx, hipFree(ctx->d_ent_work));
        ctx->d_ent_work = nullptr;
        ctx->ent_work_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_ent_work, wb));
                ctx->ent_work_bytes = wb;
    }
    // three launches, no host round trip and no copy: pack (one walk over the symbols), tile sums, place; the placing kernel
    // writes the header and puts {payload bits, error} into the host-mapped status block; nothing is written past the
    // caller's buffer
    const size_t cap_words = ((cap - 16) / 16) * 4; // whole 16-byte units behind the header
    unsigned long long total_bits = 0;
    int err = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        const int mode = (attempt == 0 && quality <= ctx->ent_lane_max_quality) ? kEntropyLanePerBlock : kEntropyEightLanes;
        const int par = ctx->ent_parity;
        ctx->ent_parity ^= 1;
        HIPCHK(ctx, entropy_gpu_fused((const int16_t *)d_coeffs_zz, n, 1, ctx->d_huff, ctx->d_ent_work, ctx->ent_work_bytes, d_out, 0,
                                      cap_words, h, w, quality, nullptr, ctx->d_stat, ctx->d_err + par, ctx->d_err + (par ^ 1), mode, ctx->stream));
        HIPCHK(ctx, wait_stream(ctx));
        total_bits = ((volatile unsigned long long *)ctx->h_stat)[0];
        err = (int)(((volatile unsigned long long *)ctx->h_stat)[1] & 0xffffffffull);
        if (err != 4 || mode == kEntropyEightLanes) break;
        ctx->ent_lane_max_quality = quality - 1; // a block of this frame needs more than a lane string holds: 8-lane kernel from here on
    }
    if (err == 1) return set_err(ctx, TIC_E_RANGE, "coefficient without a Huffman code (reference raises KeyError)");
    const size_t payload = (size_t)((total_bits + 7) / 8);
    if (err == 2 || 16 + payload > cap)
        return set_err(ctx, TIC_E_SPACE, "output buffer too small (%zu bytes needed)", 16 + (size_t)((total_bits + 31) / 32) * 4);
    *out_len = 16 + payload;
    return TIC_OK;
}

// compress() with every stage on the device: transform kernels + device entropy stage; image and stream in HBM.
int tic_compress_dev(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality, void *d_out,
                     size_t cap, size_t *out_len) {
    TIC_LOCK(ctx);
    int rc = check_stream_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    const size_t n = num_blocks(h, w);
    rc = ensure_scratch(ctx, 0, n * 128 + 16);
    if (rc) return rc;
    rc = tic_dctq_dev(ctx, d_image, h, w, row_stride, quality, ctx->d_coef, TIC_KERNEL_HYBRID);
    if (rc) return rc;
    return tic_entropy_encode_dev(ctx, ctx->d_coef, h, w, quality, d_out, cap, out_len);
}

// Asynchronous form of tic_compress_dev: the frame's launches (transform, pack, tile sums, place) are queued and the call returns; a
// caller that compresses resident frames back to back pays the submission ramp and the completion wake-up once per burst instead of once
// per frame.  Round 6: ONE context, TWO lanes.  Transform and packing of ticket t run on lane t & 1 - a stream, a coefficient buffer, an
// entropy workspace and a pair of error flags of its own - and only the placing kernel, the stage's single writer of the caller's buffer and
// of the ticket's status pair, is queued on the context's stream, behind the lane's packing and in ticket order.  The transform of frame
// t + 1 (bound by HBM) and its packing thus run beside the packing (bound by vector issue) and placing of frame t: what round 5 reached
// only from two contexts and two host threads (39 against 48 us per 4096^2 frame).  Ordering: a lane starts behind everything the context's
// stream held when the burst's first ticket was issued (every synchronous entry point returns with that stream drained, so nothing else can
// be in front of a later ticket); a lane's next frame waits for the placing of its previous one (it reuses the workspace that kernel
// reads); whatever is queued on the context's stream afterwards runs behind the placing kernels, hence behind every kernel of the tickets.
// The placing kernel leaves {payload bits, error} in the ticket's pair of the host-mapped status block; tic_async_result reads it once the
// ticket's event has fired.  Until then the caller leaves the frame's input and output alone.
int tic_compress_dev_async(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality, void *d_out, size_t cap,
                           long long *ticket) {
    TIC_LOCK(ctx);
    if (!ctx || !ticket) return TIC_E_ARG;
    int rc = check_stream_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (!d_out || cap < 16) return set_err(ctx, TIC_E_SPACE, "output buffer too small");
    if (((uintptr_t)d_out & 15u) != 0) return set_err(ctx, TIC_E_ARG, "device output buffer must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long long t = ctx->async_next;
    tic_ctx::AsyncSlot &sl = ctx->async_slots[t % kAsyncSlots];
    if (sl.ticket >= 0) return set_err(ctx, TIC_E_ARG, "%d asynchronous calls are open: collect results (tic_async_result) first", kAsyncSlots);
    if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    volatile unsigned long long *hs = ctx->h_stat + 8 + 2 * (t % kAsyncSlots);
    unsigned long long *ds = ctx->d_stat + 8 + 2 * (t % kAsyncSlots);
    const size_t n = num_blocks(h, w);
    sl.cap = cap;
    sl.empty_image = n == 0;
    if (n == 0) { // header only (codec.py:151: an empty image is a 16-byte stream)
        uint8_t hdr[16];
        write_header(hdr, h, w, quality);
        HIPCHK(ctx, hipMemcpyAsync(d_out, hdr, 16, hipMemcpyHostToDevice, ctx->stream));
    } else {
        if (!d_image) return set_err(ctx, TIC_E_ARG, "null image pointer");
        tic_ctx::AsyncLane &ln = ctx->lanes[t & 1];
        if (!ln.stream) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            HIPCHK(ctx, hipEventCreateWithFlags(&ln.packed, hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ln.placed[0], hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ln.placed[1], hipEventDisableTiming));
            HIPCHK(ctx, hipMalloc((void **)&ln.d_err, 4 * sizeof(int)));
            HIPCHK(ctx, hipMemset(ln.d_err, 0, 4 * sizeof(int)));
        }
        if (!ctx->lane_order) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->lane_order, hipEventDisableTiming));
        const size_t coef_bytes = n * 128 + 16, wb = entropy_fused_work_bytes(n);
        if (coef_bytes > ln.coef_cap || wb > ln.work_bytes) { // (grows only between bursts of one geometry: frames in flight still use the old buffers)
            HIPCHK(ctx, hipStreamSynchronize(ln.stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (coef_bytes > ln.coef_cap) {
                if (ln.d_coef) HIPCHK(ctx, hipFree(ln.d_coef));
                ln.d_coef = nullptr, ln.coef_cap = 0;
                HIPCHK(ctx, hipMalloc(&ln.d_coef, coef_bytes));
                ln.coef_cap = coef_bytes;
            }
            if (wb > ln.work_bytes) {
                for (int k = 0; k < 2; k++) {
                    if (ln.d_work[k]) HIPCHK(ctx, hipFree(ln.d_work[k]));
                    ln.d_work[k] = nullptr;
                }
                ln.work_bytes = 0;
                HIPCHK(ctx, hipMalloc(&ln.d_work[0], wb));
                HIPCHK(ctx, hipMalloc(&ln.d_work[1], wb));
                ln.work_bytes = wb;
            }
        }
        hs[0] = 0;
        hs[1] = 0;
        if (ctx->async_open == 0) { // a burst begins: its lanes start behind whatever the context's stream holds now
            HIPCHK(ctx, hipEventRecord(ctx->lane_order, ctx->stream));
            ctx->lane_epoch++;
        }
        if (ln.order_seen != ctx->lane_epoch) {
            HIPCHK(ctx, hipStreamWaitEvent(ln.stream, ctx->lane_order, 0));
            ln.order_seen = ctx->lane_epoch;
        }
        const int wk = (int)(ln.frames & 1), ek = (int)(ln.frames & 3);
        if (ln.placed_valid[wk]) HIPCHK(ctx, hipStreamWaitEvent(ln.stream, ln.placed[wk], 0)); // the frame that used this workspace two lane-frames ago has been placed
        DctqArgs a = make_args(ctx, d_image, h, w, row_stride, quality, ln.d_coef);
        HIPCHK(ctx, launch_dctq(a, dctq_kernel_id(TIC_KERNEL_HYBRID), ln.stream));
        const size_t cap_words = ((cap - 16) / 16) * 4;
        // (the 8-lane packing kernel: it takes any block the format allows, so no second run can be needed)
        HIPCHK(ctx, entropy_gpu_fused((const int16_t *)ln.d_coef, n, 1, ctx->d_huff, ln.d_work[wk], ln.work_bytes, d_out, 0, cap_words, h, w, quality, nullptr, ds,
                                      ln.d_err + ek, ln.d_err + ((ek + 3) & 3), kEntropyEightLanes, ln.stream, ctx->stream, ln.packed));
        HIPCHK(ctx, hipEventRecord(ln.placed[wk], ctx->stream));
        ln.placed_valid[wk] = true;
        ln.frames++;
    }
    HIPCHK(ctx, hipEventRecord(sl.done, ctx->stream));
    sl.ticket = t;
    ctx->async_open++;
    ctx->async_next = t + 1;
    *ticket = t;
    return TIC_OK;
}

// Result of an asynchronous call: TIC_OK and the stream's length, or that frame's error (TIC_E_RANGE: a coefficient without a Huffman
// code, TIC_E_SPACE: the stream did not fit).  wait == 0: returns TIC_E_BUSY while the frame is still in flight.  A ticket is closed
// by the call that returns anything but TIC_E_BUSY.
int tic_async_result(tic_ctx *ctx, long long ticket, int wait, size_t *out_len) {
    TIC_LOCK(ctx);
    if (!ctx || !out_len || ticket < 0) return TIC_E_ARG;
    tic_ctx::AsyncSlot &sl = ctx->async_slots[ticket % kAsyncSlots];
    if (sl.ticket != ticket) return set_err(ctx, TIC_E_ARG, "ticket %lld is not open", ticket);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {   // (any return other than TIC_E_BUSY closes the ticket - a failing event call included: the slot must not stay open for good)
        const hipError_t q = wait ? hipEventSynchronize(sl.done) : hipEventQuery(sl.done);
        if (!wait && q == hipErrorNotReady) return TIC_E_BUSY;
        if (q != hipSuccess) { sl.ticket = -1; ctx->async_open--; }
        HIPCHK(ctx, q);
    }
    sl.ticket = -1;
    ctx->async_open--;
    if (sl.empty_image) {
        *out_len = 16;
        return TIC_OK;
    }
    volatile unsigned long long *hs = ctx->h_stat + 8 + 2 * (ticket % kAsyncSlots);
    const unsigned long long total_bits = hs[0];
    const int err = (int)(hs[1] & 0xffffffffull);
    if (err == 1) return set_err(ctx, TIC_E_RANGE, "coefficient without a Huffman code (reference raises KeyError)");
    const size_t payload = (size_t)((total_bits + 7) / 8);
    if (err == 2 || 16 + payload > sl.cap)
        return set_err(ctx, TIC_E_SPACE, "output buffer too small (%zu bytes needed)", 16 + (size_t)((total_bits + 31) / 32) * 4);
    *out_len = 16 + payload;
    return TIC_OK;
}

int tic_compress(tic_ctx *ctx, const uint8_t *image, int h, int w, ptrdiff_t row_stride, int quality, uint8_t *out,
                 size_t cap, size_t *out_len) {
    TIC_LOCK(ctx);
    int rc = check_stream_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (!out || !out_len) return set_err(ctx, TIC_E_ARG, "null output pointer");
    const size_t n = num_blocks(h, w);
    if (n == 0) return entropy_encode(nullptr, h, w, quality, out, cap, out_len);
    if (!image) return set_err(ctx, TIC_E_ARG, "null image pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // image -> HBM, transform stage, entropy stage, all on the device; only the finished stream returns
    const size_t pitch = align_up((size_t)w, 256);
    rc = ensure_scratch(ctx, pitch * (size_t)h, n * 128 + 16);
    if (rc) return rc;
    const size_t need = compress_bound(h, w);
    const bool small = need <= kSmallHostBytes && !test_hook("TIC_NO_SMALL_PATH"); // the placing kernel writes the stream into host memory itself
    if (small) {
        rc = ensure_small(ctx, kSmallHostBytes);
        if (rc) return rc;
    } else if (need > ctx->d_stream_cap) {
        if (ctx->d_stream_buf) HIPCHK(ctx, hipFree(ctx->d_stream_buf));
        ctx->d_stream_buf = nullptr;
        ctx->d_stream_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_stream_buf, need));
        ctx->d_stream_cap = need;
    }
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_img, pitch, image, (size_t)row_stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice,
                                 ctx->stream));
    size_t len = 0;
    rc = tic_compress_dev(ctx, ctx->d_img, h, w, (ptrdiff_t)pitch, quality, small ? (void *)ctx->d_small : ctx->d_stream_buf,
                          small ? ctx->small_cap : ctx->d_stream_cap, &len);
    if (rc) return rc;
    if (len > cap) return set_err(ctx, TIC_E_SPACE, "output buffer too small (%zu bytes needed, %zu given)", len, cap);
    if (small) {
        memcpy(out, ctx->h_small, len); // (tic_compress_dev returned behind a drained stream: the kernel's stores have arrived)
    } else {
        HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_stream_buf, len, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out_len = len;
    return TIC_OK;
}

// ---- batch pipeline (BASELINE config 3) -------------------------------------------------------------------
// Frames travel in chunks of up to kChunk frames: one pinned staging buffer, one H2D copy, ONE kernel launch
// (grid row per frame) and one D2H copy per chunk, alternating between two streams so that the copy of chunk
// c+1 overlaps the kernel of chunk c and the read-back of chunk c-1.  Worker threads entropy-code frame by frame.

// Row pitch of staged frames: rows that are already a multiple of 8 bytes are staged back to back (one memcpy per
// frame when the caller's rows are contiguous too); other widths are padded so that 8-byte row loads stay aligned.
static inline size_t batch_pitch(int w) { return (w % 8 == 0) ? (size_t)w : align_up((size_t)w, 256); }

static void stage_frame(uint8_t *dst, size_t pitch, const uint8_t *src, ptrdiff_t row_stride, int h, int w) {
    if ((size_t)row_stride == pitch && pitch == (size_t)w) {
        memcpy(dst, src, (size_t)h * (size_t)w);
        return;
    }
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * pitch, src + (ptrdiff_t)y * row_stride, (size_t)w);
}

// Stages the frames of a chunk into the slot's pinned buffer on several threads: one thread copies ~10 GB/s, which
// would cap a 1080p batch at ~5,000 frames/s - below what PCIe and the GPU take.
static void stage_chunk(const tic_ctx *ctx, uint8_t *pin, size_t img_bytes, size_t pitch, const uint8_t *const *images, int first, int cnt,
                        ptrdiff_t row_stride, int h, int w) {
    unsigned hw = std::thread::hardware_concurrency();
    int T = ctx->stage_threads > 0 ? ctx->stage_threads : (int)(hw ? hw / 2 : 4);
    T = T < 1 ? 1 : (T > 8 ? 8 : T);
    if (T > cnt) T = cnt;
    if (T <= 1 || img_bytes * (size_t)cnt < (4u << 20)) {
        for (int k = 0; k < cnt; k++) stage_frame(pin + (size_t)k * img_bytes, pitch, images[first + k], row_stride, h, w);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([=]() {
            bind_pipeline_thread(ctx);
            for (int k = t; k < cnt; k += T) stage_frame(pin + (size_t)k * img_bytes, pitch, images[first + k], row_stride, h, w);
        });
    for (auto &x : th) x.join();
}

// Host -> device copy of a chunk.  Frames the caller holds in pinned or registered memory, rows back to back, go to the device from
// where they lie (one copy for the chunk when the frames follow each other in memory, else one per frame); anything else is staged
// into the slot's pinned buffer first (pageable memory: the runtime would stage it too, synchronously and through a small bounce
// buffer).  Returns the number of frames that took the direct path.
static bool host_pointer_is_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // (plain malloc'ed memory: "invalid value", not an error of ours)
        return false;
    }
    return at.type == hipMemoryTypeHost;
}
// Pageable input.  Copying 2 MB frames into the pinned slots costs the host 2 x the batch in DRAM traffic and eight copy threads,
// and it is what makes the host -> host rate depend on the box: 12.3 ms for 256 x 1080p on a quiet host, 18-22 ms when the copy
// threads compete with other tenants or sit on the wrong side of the socket link (profiles/r04_numa_probe.txt), while frames the
// DMA engine reads where they lie take 12.7-13.5 ms everywhere.  hipHostRegister is cheap PER CALL, not per byte - 0.03 ms for a
// chunk's 36 MB, 0.18 ms for 510 MB, the pages are validated when the copy engine first touches them - but 77 us per call: one by
// one, 256 frames cost 19.7 ms.  So the frames [first, first + cnt) are registered as ONE range, from the lowest to the highest
// address, when that range is dense enough to be mostly frames (arrays allocated one after the other are 16 bytes to a few pages
// apart) - for the whole batch if possible, else chunk by chunk.  Anything that cannot be registered this way (a range with a
// hole, memory of another kind, part of it registered by the caller already) is staged as before.  Returns true if the frames
// are now pinned.
static bool auto_register_frames(tic_ctx *ctx, const uint8_t *const *images, int first, int cnt, size_t img_bytes) {
    if (!ctx->auto_register || cnt < 1 || img_bytes < (256u << 10)) return false;
    // (a range over frames of which some are pinned already would overlap the caller's own registration: such a set is left alone)
    for (int k = 0; k < cnt; k++)
        if (host_pointer_is_pinned(images[first + k]) || host_pointer_is_pinned(images[first + k] + img_bytes - 1)) return false;
    uintptr_t lo = UINTPTR_MAX, hi = 0;
    for (int k = 0; k < cnt; k++) {
        const uintptr_t p = (uintptr_t)images[first + k];
        if (!p) return false;
        lo = p < lo ? p : lo;
        hi = p + img_bytes > hi ? p + img_bytes : hi;
    }
    lo &= ~(uintptr_t)4095;
    hi = (hi + 4095) & ~(uintptr_t)4095;
    const size_t span = hi - lo, frames = img_bytes * (size_t)cnt;
    if (span > frames + frames / 4 + (1u << 20)) return false; // the frames lie scattered: a range over them would pin memory that is not theirs
    if (hipHostRegister((void *)lo, span, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    ctx->autoregs.push_back((void *)lo);
    ctx->last_batch_autoreg_frames += cnt;
    return true;
}
static void auto_unregister_all(tic_ctx *ctx) { // (after the call's last copy has completed)
    for (void *r : ctx->autoregs) {
        if (hipHostUnregister(r) != hipSuccess) (void)hipGetLastError();
    }
    ctx->autoregs.clear();
}

static hipError_t upload_chunk(tic_ctx *ctx, Slot &s, size_t img_bytes, size_t pitch, const uint8_t *const *images, int first, int cnt,
                               ptrdiff_t row_stride, int h, int w, hipStream_t st, int *direct) {
    *direct = 0;
    const bool dense = (size_t)row_stride == pitch && pitch == (size_t)w; // the device layout IS the caller's layout
    bool pinned = dense;
    for (int k = 0; k < cnt && pinned; k++)
        pinned = host_pointer_is_pinned(images[first + k]) && host_pointer_is_pinned(images[first + k] + img_bytes - 1);
    if (!pinned && dense) {
        // a chunk of pageable frames (chunk by chunk: the whole batch was tried at the call's start)
        BT_START();
        pinned = auto_register_frames(ctx, images, first, cnt, img_bytes);
        BT_STOP(0);
    }
    if (pinned) {
        // from where the frames lie: one copy for the chunk when they follow each other in memory, else one per frame.  A copy the
        // runtime refuses (frames of one chunk in two separately registered ranges make a chunk-wide copy invalid) falls back to
        // one copy per frame, and that to the staged route below - the device buffer is simply written again, in stream order
        bool contiguous = true;
        for (int k = 1; k < cnt && contiguous; k++) contiguous = images[first + k] == images[first] + (size_t)k * img_bytes;
        hipError_t e = contiguous ? hipMemcpyAsync(s.d_img, images[first], img_bytes * cnt, hipMemcpyHostToDevice, st) : hipErrorInvalidValue;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipSuccess;
            for (int k = 0; k < cnt && e == hipSuccess; k++)
                e = hipMemcpyAsync((char *)s.d_img + (size_t)k * img_bytes, images[first + k], img_bytes, hipMemcpyHostToDevice, st);
        }
        if (e == hipSuccess) {
            *direct = cnt;
            return hipSuccess;
        }
        (void)hipGetLastError();
    }
    BT_START();
    stage_chunk(ctx, s.pin_in, img_bytes, pitch, images, first, cnt, row_stride, h, w);
    BT_STOP(0);
    return hipMemcpyAsync(s.d_img, s.pin_in, img_bytes * cnt, hipMemcpyHostToDevice, st);
}

static int ensure_batch_slots(tic_ctx *ctx, int h, int w, int chunk) {
    const int S = 4; // staging, device, read-back, hand-out: one chunk each
    const size_t nblk = num_blocks(h, w);
    const size_t pitch = batch_pitch(w);
    const size_t img_bytes = pitch * (size_t)h, coef_bytes = nblk * 128;
    const size_t need_img = img_bytes * chunk, need_coef = coef_bytes * chunk;
    if ((int)ctx->bslots.size() != S || ctx->bslot_h != h || ctx->bslot_w != w || ctx->bslot_chunk < chunk) {
        for (auto &sl : ctx->bslots) {
            if (sl.pin_in) (void)hipHostFree(sl.pin_in);
            if (sl.pin_out) (void)hipHostFree(sl.pin_out);
            if (sl.d_img) (void)hipFree(sl.d_img);
            if (sl.d_coef) (void)hipFree(sl.d_coef);
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.rb_done) (void)hipEventDestroy(sl.rb_done);
            if (sl.d_work) (void)hipFree(sl.d_work);
            if (sl.d_lens) (void)hipFree(sl.d_lens);
            if (sl.h_lens) (void)hipHostFree(sl.h_lens);
            if (sl.d_err) (void)hipFree(sl.d_err);
            if (sl.h_err) (void)hipHostFree(sl.h_err);
            if (sl.d_streams) (void)hipFree(sl.d_streams);
        }
        ctx->bslots.assign(S, Slot());
        ctx->bslot_img_bytes = ctx->bslot_coef_bytes = 0;
        ctx->bslot_h = ctx->bslot_w = -1;
        ctx->bslot_chunk = 0;
        for (auto &sl : ctx->bslots) {
            hipError_t e;
            if ((e = hipHostMalloc((void **)&sl.pin_in, need_img, hipHostMallocDefault)) != hipSuccess ||
                (e = hipHostMalloc((void **)&sl.pin_out, need_coef, hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc(&sl.d_img, need_img)) != hipSuccess || (e = hipMalloc(&sl.d_coef, need_coef)) != hipSuccess ||
                (e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)) != hipSuccess ||
                (e = hipEventCreateWithFlags(&sl.rb_done, hipEventDisableTiming)) != hipSuccess)
                return set_err(ctx, TIC_E_HIP, "batch buffer allocation failed: %s", hipGetErrorString(e));
            sl.work_bytes = entropy_fused_work_bytes((nblk + 8) * (size_t)chunk); // (every frame's partitions are rounded up)
            sl.parity = 0;
            if ((e = hipMalloc(&sl.d_work, sl.work_bytes)) != hipSuccess ||
                (e = hipMalloc((void **)&sl.d_lens, chunk * sizeof(unsigned long long))) != hipSuccess ||
                (e = hipHostMalloc((void **)&sl.h_lens, chunk * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc((void **)&sl.d_err, 2 * sizeof(int))) != hipSuccess || (e = hipMemset(sl.d_err, 0, 2 * sizeof(int))) != hipSuccess ||
                (e = hipHostMalloc((void **)&sl.h_err, sizeof(int), hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc(&sl.d_streams, align_up(compress_bound(h, w), 16) * (size_t)chunk)) != hipSuccess)
                return set_err(ctx, TIC_E_HIP, "batch entropy workspace allocation failed: %s", hipGetErrorString(e));
        }
        ctx->bslot_img_bytes = need_img;
        ctx->bslot_coef_bytes = need_coef;
        ctx->bslot_h = h;
        ctx->bslot_w = w;
        ctx->bslot_chunk = chunk;
    }
    return TIC_OK;
}

static int batch_impl(tic_ctx *ctx, const uint8_t *const *images, int n, int h, int w, ptrdiff_t row_stride, int quality,
                      int16_t *const *coeffs, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int threads,
                      bool want_entropy) {
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !images)) return set_err(ctx, TIC_E_ARG, "bad batch arguments");
    if (want_entropy && n > 0 && (!outs || !caps || !out_lens)) return set_err(ctx, TIC_E_ARG, "null output arrays");
    if (n == 0) return TIC_OK;
    const size_t nblk = num_blocks(h, w);
    if (nblk == 0) {
        for (int i = 0; i < n && want_entropy; i++) {
            int r = entropy_encode(nullptr, h, w, quality, outs[i], caps[i], &out_lens[i]);
            if (r) return r;
        }
        return TIC_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool need_d2h = want_entropy || coeffs != nullptr;
    const size_t pitch = batch_pitch(w);
    const size_t img_bytes = pitch * (size_t)h, coef_bytes = nblk * 128;
    const int chunk = chunk_frames(n, img_bytes);
    const int S = 3;
    int result = TIC_OK;
    rc = ensure_batch_slots(ctx, h, w, chunk);
    if (rc) return rc;
    std::vector<Slot> &slots = ctx->bslots;
    for (auto &sl : slots) sl.remaining = 0;
    auto cleanup = [&]() {};

    std::mutex mu;
    std::condition_variable cv_job, cv_free;
    std::deque<std::pair<int, int>> jobs; // (slot, frame index inside the chunk)
    bool closing = false;
    std::atomic<int> first_err{TIC_OK};

    ctx->last_batch_direct_frames = ctx->last_batch_staged_frames = ctx->last_batch_autoreg_frames = 0;
    ctx->bt = BatchTrace();
    if ((size_t)row_stride == pitch && pitch == (size_t)w) { // the whole batch as one range, if it is one
        BT_START();
        (void)auto_register_frames(ctx, images, 0, n, img_bytes);
        BT_STOP(0);
    }
    auto consumer = [&]() {
        bind_pipeline_thread(ctx);
        (void)hipSetDevice(ctx->device);
        for (;;) {
            std::pair<int, int> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return !jobs.empty() || closing; });
                if (jobs.empty()) return;
                job = jobs.front();
                jobs.pop_front();
            }
            Slot &s = slots[job.first];
            const int f = s.first + job.second;
            int r = hipEventSynchronize(s.done) == hipSuccess ? TIC_OK : TIC_E_HIP;
            const int16_t *zz = need_d2h ? s.pin_out + (size_t)job.second * nblk * 64 : nullptr;
            if (r == TIC_OK && want_entropy) r = entropy_encode(zz, h, w, quality, outs[f], caps[f], &out_lens[f]);
            if (r == TIC_OK && coeffs && coeffs[f]) memcpy(coeffs[f], zz, coef_bytes);
            if (r != TIC_OK) {
                int exp = TIC_OK;
                first_err.compare_exchange_strong(exp, r);
            }
            bool freed;
            {
                std::lock_guard<std::mutex> lk(mu);
                freed = --s.remaining == 0;
            }
            if (freed) cv_free.notify_all();
        }
    };
    int nthreads = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(consumer);

    int c = 0;
    for (int first = 0; first < n && result == TIC_OK; first += chunk, c++) {
        const int cnt = n - first < chunk ? n - first : chunk;
        const int si = c % S;
        Slot &s = slots[si];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_free.wait(lk, [&] { return s.remaining == 0; });
            s.first = first;
            s.count = cnt;
            s.remaining = cnt;
        }
        hipStream_t st = ctx->bstream[c & 1];
        int direct = 0;
        hipError_t e = upload_chunk(ctx, s, img_bytes, pitch, images, first, cnt, row_stride, h, w, st, &direct);
        ctx->last_batch_direct_frames += direct;
        ctx->last_batch_staged_frames += cnt - direct;
        BT_START();
        if (e == hipSuccess) {
            DctqArgs a = make_args(ctx, s.d_img, h, w, (ptrdiff_t)pitch, quality, s.d_coef);
            a.fallback_count = nullptr;
            a.nframes = cnt;
            a.frame_stride_in = (long)img_bytes;
            a.frame_stride_out = (long)coef_bytes;
            merge_frames(a);
            e = launch_dctq(a, 2, st);
        }
        if (e == hipSuccess && need_d2h) e = hipMemcpyAsync(s.pin_out, s.d_coef, coef_bytes * cnt, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(s.done, st);
        BT_STOP(1);
        if (e != hipSuccess) {
            result = set_err(ctx, TIC_E_HIP, "batch enqueue failed at frame %d: %s", first, hipGetErrorString(e));
            std::lock_guard<std::mutex> lk(mu);
            s.remaining = 0;
            break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            for (int k = 0; k < cnt; k++) jobs.emplace_back(si, k);
        }
        cv_job.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        closing = true;
    }
    cv_job.notify_all();
    for (auto &t : pool) t.join();
    (void)hipStreamSynchronize(ctx->bstream[0]);
    (void)hipStreamSynchronize(ctx->bstream[1]);
    auto_unregister_all(ctx);
    cleanup();
    if (result == TIC_OK && first_err.load() != TIC_OK)
        result = set_err(ctx, first_err.load(), "batch consumer failed with code %d", first_err.load());
    return result;
}

// Batch compress with the entropy stage on the device: per chunk one H2D copy, one transform launch, the three
// entropy steps, then only the finished streams (and 8 bytes of length per frame) come back.
constexpr int kRetryEightLanes = -1000; // internal: compress_batch_gpu asks tic_compress_batch for another run with the 8-lane packing kernel

// Read-back of a chunk's streams by the SHADER, not by a DMA engine: rows of `row16` 16-byte pieces from device memory into the slot's
// pinned host buffer (device-accessible).  The runtime spreads the batch's uploads over both SDMA engines and queues every later copy
// behind the uploads already submitted - with four chunks of uploads in the queue a chunk's read-back started 2 ms after its kernels
// had finished, the read-backs came in bursts of four, and the uploads stalled 0.5-0.7 ms behind every burst for want of a free slot
// (rocprofv3 --memory-copy-trace, profiles/r04_batch_timeline.txt).  Stores from a kernel cross the link in the other direction while
// the engines upload.
__global__ __launch_bounds__(256) void readback_rows_kernel(const uint4 *__restrict__ src, size_t src_pitch16, uint4 *__restrict__ dst, size_t dst_pitch16,
                                                            size_t row16) {
    const uint4 *s = src + (size_t)blockIdx.y * src_pitch16;
    uint4 *d = dst + (size_t)blockIdx.y * dst_pitch16;
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < row16; i += (size_t)gridDim.x * 256u)
        __builtin_nontemporal_store(reinterpret_cast<const u32x4v *>(s)[i], reinterpret_cast<u32x4v *>(d) + i);
}

// ... and straight into the CALLER's buffers when those are rows of one block of memory (a pool of n x cap bytes: what compress_batch() of the
// Python mirror and bench.py hold), pinned for the call: 8-byte pieces, because a row pitch of tic_compress_bound() bytes is a multiple of 8, not 16.
__global__ __launch_bounds__(256) void readback_rows8_kernel(const uint2 *__restrict__ src, size_t src_pitch8, uint2 *__restrict__ dst, size_t dst_pitch8, size_t row8) {
    const uint2 *s = src + (size_t)blockIdx.y * src_pitch8;
    uint2 *d = dst + (size_t)blockIdx.y * dst_pitch8;
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < row8; i += (size_t)gridDim.x * 256u)
        __builtin_nontemporal_store(reinterpret_cast<const u32x2v *>(s)[i], reinterpret_cast<u32x2v *>(d) + i);
}

static int compress_batch_gpu(tic_ctx *ctx, const uint8_t *const *images, int n, int h, int w, ptrdiff_t row_stride,
                              int quality, uint8_t *const *outs, const size_t *caps, size_t *out_lens) {
    int rc = check_stream_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!images || !outs || !caps || !out_lens))) return set_err(ctx, TIC_E_ARG, "bad batch arguments");
    if (n == 0) return TIC_OK;
    const size_t nblk = num_blocks(h, w);
    if (nblk == 0) {
        for (int i = 0; i < n; i++) {
            int r = entropy_encode(nullptr, h, w, quality, outs[i], caps[i], &out_lens[i]);
            if (r) return r;
        }
        return TIC_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t pitch = batch_pitch(w);
    const size_t img_bytes = pitch * (size_t)h, coef_bytes = nblk * 128, bound = align_up(compress_bound(h, w), 16);
    const int chunk = chunk_frames(n, img_bytes);
    rc = ensure_batch_slots(ctx, h, w, chunk);
    if (rc) return rc;
    std::vector<Slot> &slots = ctx->bslots;
    const int S = (int)slots.size();
    int result = TIC_OK;
    const bool inline_path = n <= chunk; // a batch of ONE chunk runs on the calling thread (below)
    // finishing a chunk, part 1: wait for its lengths, then read its streams back into pinned memory
    auto read_back = [&](Slot &s, hipStream_t st) -> int {
        if (s.count == 0) return TIC_OK;
        BT_START();
        const hipError_t ev = hipEventSynchronize(s.done);
        BT_STOP(2);
        if (ev != hipSuccess) return set_err(ctx, TIC_E_HIP, "batch chunk failed");
        if (*s.h_err == 1) return set_err(ctx, TIC_E_RANGE, "coefficient without a Huffman code (reference raises KeyError)");
        if (*s.h_err == 4) return kRetryEightLanes; // a block exceeds the lane-per-block kernel's strings: the whole call is run again
        if (*s.h_err) return set_err(ctx, TIC_E_SPACE, "device entropy stage: stream buffer too small");
        // The streams come back into the slot's pinned buffer (asynchronous DMA; a copy straight into the caller's pageable
        // buffers is staged by the runtime, ~0.15 ms each) and are handed out by a few threads.
        size_t maxlen = 0;
        for (int k = 0; k < s.count; k++) {
            const size_t len = (size_t)s.h_lens[k];
            const int f = s.first + k;
            if (len > caps[f]) return set_err(ctx, TIC_E_SPACE, "output buffer of frame %d too small (%zu bytes needed)", f, len);
            out_lens[f] = len;
            if (len > maxlen) maxlen = len;
        }
        // ONE strided copy brings the head of every frame's stream buffer - as many bytes as the longest stream has - into
        // the slot's pinned buffer (16 separate copies of ~0.9 MB cost ~45 us each, 2.5 x their transfer time); frames of one
        // batch compress to similar sizes, so little more than the streams themselves crosses PCIe.
        // Zero copy (round 6; a batch of one chunk only - it runs on the calling thread, which owns the call's registrations): the caller's
        // buffers are rows of ONE block of memory (equal distances, 8-byte aligned: a pool of n x cap bytes) -> the block is pinned for the
        // call and the shader stores every stream where the caller wants it; the copy out of the pipeline's pinned buffer (0.12-0.28 ms for the
        // 49 streams of the reference's benchmark set: a third of the call) does not happen.  The rows are written up to the chunk's longest
        // stream rounded to 8 bytes (all within the frames' capacities, checked): bytes behind a stream's end are not preserved.
        if (inline_path && ctx->auto_register && s.count >= 2 && maxlen >= 16) {
            const uint8_t *base = outs[s.first];
            const size_t P = (size_t)(outs[s.first + 1] - outs[s.first]), row8 = (maxlen + 7) / 8;
            bool ok = outs[s.first + 1] > outs[s.first] && P % 8 == 0 && (uintptr_t)base % 8 == 0 && P >= row8 * 8;
            for (int k = 0; k < s.count && ok; k++) ok = outs[s.first + k] == base + (size_t)k * P && caps[s.first + k] >= row8 * 8;
            const size_t span = ok ? (size_t)(s.count - 1) * P + row8 * 8 : 0;
            void *reg = nullptr;
            if (ok && !(host_pointer_is_pinned(base) && host_pointer_is_pinned(base + span - 1))) {
                const uintptr_t lo = (uintptr_t)base & ~(uintptr_t)4095, hi = ((uintptr_t)base + span + 4095) & ~(uintptr_t)4095;
                if (hipHostRegister((void *)lo, hi - lo, hipHostRegisterDefault) == hipSuccess) {
                    reg = (void *)lo;
                    ctx->autoregs.push_back(reg); // (released with the call's other registrations, behind its last synchronisation)
                } else {
                    (void)hipGetLastError();
                    ok = false;
                }
            }
            void *d_dst = nullptr;
            if (ok && hipHostGetDevicePointer(&d_dst, (void *)base, 0) != hipSuccess) {
                (void)hipGetLastError();
                ok = false;
            }
            if (ok) {
                BT_START();
                hipLaunchKernelGGL(readback_rows8_kernel, dim3(64, (unsigned)s.count), dim3(256), 0, st, (const uint2 *)s.d_streams, bound / 8, (uint2 *)d_dst, P / 8, row8);
                hipError_t e = hipGetLastError();
                if (e == hipSuccess) e = hipEventRecord(s.rb_done, st);
                BT_STOP(3);
                if (e != hipSuccess) return set_err(ctx, TIC_E_HIP, "stream read-back failed: %s", hipGetErrorString(e));
                s.rb_row = 0; // (nothing to hand out)
                ctx->last_batch_zero_copy += s.count;
                return TIC_OK;
            }
        }
        const size_t pin_cap = coef_bytes * (size_t)chunk; // size of pin_out (ensure_batch_slots)
        const size_t row = align_up(maxlen, 256);
        const bool packed = row * (size_t)s.count <= pin_cap;
        BT_START();
        if (packed) {
            // (round 3: ONE strided hipMemcpy2DAsync per chunk - 16 separate copies of ~0.9 MB cost ~45 us each; now the kernel above)
            const size_t row16 = (maxlen + 15) / 16; // (row and bound are multiples of 16; the bytes behind a stream are never handed out)
            hipLaunchKernelGGL(readback_rows_kernel, dim3(64, (unsigned)s.count), dim3(256), 0, st, (const uint4 *)s.d_streams, bound / 16, (uint4 *)s.pin_out, row / 16,
                               row16);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return set_err(ctx, TIC_E_HIP, "stream read-back failed: %s", hipGetErrorString(e));
        } else {
            for (int k = 0; k < s.count; k++) {
                hipError_t e = hipMemcpyAsync(outs[s.first + k], (char *)s.d_streams + (size_t)k * bound, (size_t)s.h_lens[k],
                                              hipMemcpyDeviceToHost, st);
                if (e != hipSuccess) return set_err(ctx, TIC_E_HIP, "stream read-back failed: %s", hipGetErrorString(e));
            }
        }
        // (the reading thread does not wait for the copy: it records an event behind it and turns to the next chunk, so that the
        // read-backs follow each other on their stream without a host round trip in between; the hand-out thread waits for the event)
        const hipError_t rb = hipEventRecord(s.rb_done, st);
        BT_STOP(3);
        if (rb != hipSuccess) return set_err(ctx, TIC_E_HIP, "stream read-back failed");
        s.rb_row = packed ? row : 0;
        return TIC_OK;
    };
    // part 2: out of pinned memory into the caller's buffers, by a few threads
    auto hand_out_chunk = [&](Slot &s) -> int {
        if (s.count == 0) return TIC_OK;
        const size_t row = s.rb_row;
        const bool packed = row != 0;
        BT_START();
        if (hipEventSynchronize(s.rb_done) != hipSuccess) return set_err(ctx, TIC_E_HIP, "stream read-back failed");
        if (packed) {
            const int cnt = s.count, first = s.first;
            const char *src = (const char *)s.pin_out;
            const unsigned long long *lens = s.h_lens;
            auto hand_out = [=](int t, int T) {
                if (T > 1) bind_pipeline_thread(ctx);
                for (int k = t; k < cnt; k += T) memcpy(outs[first + k], src + (size_t)k * row, (size_t)lens[k]);
            };
            size_t total = 0;
            for (int k = 0; k < cnt; k++) total += (size_t)lens[k];
            const int T = cnt < 4 || total < (4u << 20) ? 1 : 4; // (starting and joining four threads costs 0.15-0.2 ms: more than copying 4 MB)
            if (T == 1) {
                hand_out(0, 1);
            } else {
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back(hand_out, t, T);
                for (auto &x : th) x.join();
            }
        }
        BT_STOP(4);
        s.count = 0;
        return TIC_OK;
    };
    for (auto &sl : slots) sl.count = 0;
    // Three host threads, four slots: this thread stages chunk c into pinned memory and enqueues it (H2D, kernels, lengths),
    // a second waits for each chunk in turn and reads its streams back into pinned memory, a third hands them out to the
    // caller's buffers.  The device always has work queued while the threads copy, and a slot is staged into again only after
    // the third thread has released it.
    // (History, 256 x 1080p host -> host: one thread that finished chunk c - 3 on the stream already holding chunk c - 1,
    // then staged chunk c: 28 ms, the device idle during the host copies; read-back on its own stream after enqueueing
    // chunk c: 23.5 ms; one strided read-back copy per chunk instead of 16: 21.5 ms; read-back + hand-out on a second
    // thread: 15 ms; this: see DESIGN.md section 6.)
    std::mutex mu;
    std::condition_variable cv_read, cv_hand, cv_free;
    std::deque<int> q_read, q_hand; // slots to read back / to hand out, in submission order
    std::vector<char> busy(S, 0);   // slot submitted and not yet released by the hand-out thread
    bool stop_read = false, stop_hand = false;
    int fin_result = TIC_OK;
    ctx->last_batch_direct_frames = ctx->last_batch_staged_frames = ctx->last_batch_autoreg_frames = 0;
    ctx->last_batch_zero_copy = 0;
    ctx->bt = BatchTrace();
    if ((size_t)row_stride == pitch && pitch == (size_t)w) { // the whole batch as one range, if it is one
        BT_START();
        (void)auto_register_frames(ctx, images, 0, n, img_bytes);
        BT_STOP(0);
    }
    // A batch of ONE chunk (the reference's benchmark set: 49 frames of 512 x 512) runs on the calling thread: enqueue, wait, read back, hand out.
    // Starting the two pipeline threads costs more than the chunk's work when the host is busy - their first wake-up came 3-10 ms late in
    // one call in three on a shared box (tools/batch_small_probe.py: chunk_wait 0.008 ms, the reader found the chunk long finished).
    std::thread reader, hander;
    if (!inline_path) {
    reader = std::thread([&]() {
        bind_pipeline_thread(ctx);
        (void)hipSetDevice(ctx->device);
        for (;;) {
            int k;
            {
                std::unique_lock<std::mutex> l(mu);
                cv_read.wait(l, [&] { return stop_read || !q_read.empty(); });
                if (q_read.empty()) break;
                k = q_read.front();
                q_read.pop_front();
            }
            const int r = read_back(slots[k], ctx->rstream);
            {
                std::lock_guard<std::mutex> l(mu);
                if (r != TIC_OK) {
                    if (fin_result == TIC_OK) fin_result = r;
                    slots[k].count = 0; // nothing to hand out
                }
                q_hand.push_back(k);
            }
            cv_hand.notify_one();
        }
        {
            std::lock_guard<std::mutex> l(mu);
            stop_hand = true;
        }
        cv_hand.notify_one();
    });
    hander = std::thread([&]() {
        bind_pipeline_thread(ctx);
        for (;;) {
            int k;
            {
                std::unique_lock<std::mutex> l(mu);
                cv_hand.wait(l, [&] { return stop_hand || !q_hand.empty(); });
                if (q_hand.empty()) return;
                k = q_hand.front();
                q_hand.pop_front();
            }
            const int r = hand_out_chunk(slots[k]);
            {
                std::lock_guard<std::mutex> l(mu);
                if (r != TIC_OK && fin_result == TIC_OK) fin_result = r;
                slots[k].count = 0;
                busy[k] = 0;
            }
            cv_free.notify_all();
        }
    });
    }
    int c = 0;
    for (int first = 0; first < n && result == TIC_OK; first += chunk, c++) {
        const int cnt = n - first < chunk ? n - first : chunk;
        const int si = c % S;
        Slot &s = slots[si];
        hipStream_t st = ctx->bstream[c & 1];
        {
            BT_START();
            std::unique_lock<std::mutex> l(mu);
            cv_free.wait(l, [&] { return !busy[si]; });
            BT_STOP(5);
            if (fin_result != TIC_OK) break;
        }
        s.first = first;
        s.count = cnt;
        int direct = 0;
        hipError_t e = upload_chunk(ctx, s, img_bytes, pitch, images, first, cnt, row_stride, h, w, st, &direct);
        ctx->last_batch_direct_frames += direct;
        ctx->last_batch_staged_frames += cnt - direct;
        BT_START();
        if (e == hipSuccess) {
            DctqArgs a = make_args(ctx, s.d_img, h, w, (ptrdiff_t)pitch, quality, s.d_coef);
            a.fallback_count = nullptr;
            a.nframes = cnt;
            a.frame_stride_in = (long)img_bytes;
            a.frame_stride_out = (long)coef_bytes;
            merge_frames(a);
            e = launch_dctq(a, 2, st);
        }
        const int par = s.parity;
        s.parity ^= 1;
        if (e == hipSuccess) // entropy stage of the whole chunk: pack + place (headers, lengths); no zero fill
            e = entropy_gpu_fused((const int16_t *)s.d_coef, nblk, cnt, ctx->d_huff, s.d_work, s.work_bytes, s.d_streams, bound,
                                  (bound - 16) / 4, h, w, quality, s.d_lens, nullptr, s.d_err + par, s.d_err + (par ^ 1),
                                  quality <= ctx->ent_lane_max_quality ? kEntropyLanePerBlock : kEntropyEightLanes, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s.h_lens, s.d_lens, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s.h_err, s.d_err + par, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(s.done, st);
        BT_STOP(1);
        if (e != hipSuccess) {
            result = set_err(ctx, TIC_E_HIP, "batch enqueue failed at frame %d: %s", first, hipGetErrorString(e));
            s.count = 0;
            break;
        }
        if (inline_path) {
            int r = read_back(s, ctx->rstream);
            if (r == TIC_OK) r = hand_out_chunk(s);
            s.count = 0;
            if (r != TIC_OK) result = r;
            continue;
        }
        {
            std::lock_guard<std::mutex> l(mu);
            busy[si] = 1;
            q_read.push_back(si);
        }
        cv_read.notify_one();
    }
    {
        std::lock_guard<std::mutex> l(mu);
        stop_read = true;
    }
    cv_read.notify_one();
    BT_START();
    if (reader.joinable()) reader.join(); // (each drains its queue first)
    if (hander.joinable()) hander.join();
    if (result == TIC_OK) result = fin_result;
    (void)hipStreamSynchronize(ctx->bstream[0]);
    (void)hipStreamSynchronize(ctx->bstream[1]);
    (void)hipStreamSynchronize(ctx->rstream);
    BT_STOP(6); // (tic_last_batch_phases [6]: the caller's wait for the pipeline's threads and streams, [7]: releasing the frames pinned for the call)
    BT_START();
    auto_unregister_all(ctx);
    BT_STOP(7);
    for (auto &sl : slots) sl.count = 0;
    return result;
}

int tic_compress_batch(tic_ctx *ctx, const uint8_t *const *images, int n, int h, int w, ptrdiff_t row_stride, int quality,
                       uint8_t *const *outs, const size_t *caps, size_t *out_lens, int threads) {
    TIC_LOCK(ctx);
    if (threads <= 0) {
        int rc = compress_batch_gpu(ctx, images, n, h, w, row_stride, quality, outs, caps, out_lens);
        if (rc == kRetryEightLanes) {
            ctx->ent_lane_max_quality = quality - 1;
            rc = compress_batch_gpu(ctx, images, n, h, w, row_stride, quality, outs, caps, out_lens);
        }
        return rc;
    }
    return batch_impl(ctx, images, n, h, w, row_stride, quality, nullptr, outs, caps, out_lens, threads, true);
}

// One batch over several contexts - normally one per GPU of the node - from ONE process: a host thread per context, contiguous shards
// (frame i of n goes to context i / ceil(n / nctx): the partition of DESIGN.md 7 and of tinyimgcodec_amd/distributed.py
// shard_range), every thread runs the stream-overlapped pipeline of tic_compress_batch on its shard.  Sizes come back in frame
// order in out_lens; there is no data-path exchange between the shards.  The first failing shard's code is returned (its message:
// tic_last_error of that context); *failed_ctx (may be null) receives its index, -1 when all succeeded.  Two contexts may sit on the
// same device (each has its own streams, slots and scratch): that is how a one-GPU box tests this path.
int tic_compress_batch_multi(tic_ctx *const *ctxs, int nctx, const uint8_t *const *images, int n, int h, int w, ptrdiff_t row_stride,
                             int quality, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int threads, int *failed_ctx) {
    if (failed_ctx) *failed_ctx = -1;
    if (!ctxs || nctx < 1 || n < 0) return TIC_E_ARG;
    for (int k = 0; k < nctx; k++) {
        if (!ctxs[k]) return TIC_E_ARG;
        for (int j = 0; j < k; j++)
            if (ctxs[j] == ctxs[k]) return set_err(ctxs[k], TIC_E_ARG, "context %d and %d of a multi-context batch are the same", j, k);
    }
    if (n == 0) return TIC_OK;
    if (!images || !outs || !caps || !out_lens) return set_err(ctxs[0], TIC_E_ARG, "null pointer argument");
    const int per = (n + nctx - 1) / nctx;
    std::vector<int> rcs((size_t)nctx, TIC_OK);
    std::vector<std::thread> th;
    for (int k = 0; k < nctx; k++) {
        const int lo = std::min(k * per, n), hi = std::min(lo + per, n);
        if (hi <= lo) continue;
        th.emplace_back([=, &rcs] { rcs[(size_t)k] = tic_compress_batch(ctxs[k], images + lo, hi - lo, h, w, row_stride, quality, outs + lo, caps + lo, out_lens + lo, threads); });
    }
    for (auto &t : th) t.join();
    for (int k = 0; k < nctx; k++)
        if (rcs[(size_t)k] != TIC_OK) {
            if (failed_ctx) *failed_ctx = k;
            return rcs[(size_t)k];
        }
    return TIC_OK;
}

int tic_dctq_batch(tic_ctx *ctx, const uint8_t *const *images, int n, int h, int w, ptrdiff_t row_stride, int quality,
                   int16_t *const *coeffs) {
    TIC_LOCK(ctx);
    return batch_impl(ctx, images, n, h, w, row_stride, quality, coeffs, nullptr, nullptr, nullptr, 4, false);
}

// ---- decode ---------------------------------------------------------------------------------------------
// Inverse stage on coefficients that already sit in ctx->d_coef (int16 [N][64] zig-zag, DC integrated) -> pixels in `out`.
// scaled_exp < 0: decode() proper; >= 0: its scaled_dct branch with 2 ** scaled_exp (codec.py:59-62)
static int idct_from_device(tic_ctx *ctx, int h, int w, int quality, int scaled_exp, uint8_t *out, bool out_on_device = false,
                            size_t out_stride = 0) {
    const size_t pitch = align_up((size_t)w, 256);
    IdctArgs a;
    a.coeffs = (const int16_t *)ctx->d_coef;
    a.out = (uint8_t *)ctx->d_img;
    a.h = h;
    a.w = w;
    a.stride = (long)pitch;
    a.bw = (w + 7) / 8;
    a.tiles_x = (a.bw + 7) / 8;
    a.ntiles = ((h + 7) / 8) * a.tiles_x;
    a.aligned8 = 1;
    a.consts = ctx->d_consts + (scaled_exp >= 0 ? 50 : quality); // codec.py:62: quality = 50 on the scaled branch
    a.scaled = scaled_exp >= 0;
    a.pow2 = scaled_exp >= 0 ? ldexp(1.0, scaled_exp) : 1.0;
    // a device destination whose rows are 8-byte aligned takes the pixels straight from the kernel (its row stores are cropped to w)
    const bool direct = out_on_device && out_stride % 8 == 0 && (uintptr_t)out % 8 == 0;
    if (direct) {
        a.out = out;
        a.stride = (long)out_stride;
    }
    HIPCHK(ctx, launch_idct(a, ctx->stream));
    if (!direct)
        HIPCHK(ctx, hipMemcpy2DAsync(out, out_on_device ? out_stride : (size_t)w, ctx->d_img, pitch, (size_t)w, (size_t)h,
                                     out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TIC_OK;
}

static int idctq_impl(tic_ctx *ctx, const int16_t *coeffs_zz, int h, int w, int quality, int scaled_exp, uint8_t *out, size_t cap) {
    if (!ctx) return TIC_E_ARG;
    if (h < 0 || w < 0) return set_err(ctx, TIC_E_ARG, "negative image size");
    if (scaled_exp < 0 && (quality < 1 || quality > 99) && !(quality == TIC_QUALITY_CUSTOM && ctx->custom_quality != 0.0))
        return set_err(ctx, TIC_E_QUALITY, "quality %d outside 1..99", quality);
    if (scaled_exp > 62) return set_err(ctx, TIC_E_QUALITY, "scaled_dct exponent %d outside 0..62", scaled_exp);
    const size_t n = num_blocks(h, w);
    if (n == 0) return TIC_OK;
    if (!coeffs_zz) return set_err(ctx, TIC_E_ARG, "null coefficient pointer");
    if (!out || (size_t)h * (size_t)w > cap) return set_err(ctx, TIC_E_SPACE, "output buffer too small");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t pitch = align_up((size_t)w, 256);
    int rc = ensure_scratch(ctx, pitch * (size_t)h, n * 128);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_coef, coeffs_zz, n * 128, hipMemcpyHostToDevice, ctx->stream));
    return idct_from_device(ctx, h, w, quality, scaled_exp, out);
}

// Long streams: Huffman + run-length decode AND the inverse stage on the device, fused (tic_entropy_dec_gpu.hip): the stream goes
// in, pixels come out, no coefficient array in between.  `out` / `out_on_device` / `out_stride` as idct_from_device.  Returns TIC_OK
// with *done = true when the image is complete in `out`; *done = false (and TIC_OK) when the device decoder met something unusual
// or does not apply - the caller then decodes on the host, which reproduces the reference's behaviour on malformed streams (what
// the device wrote to `out` until then is overwritten).
// Streams the device decoder takes.  Long ones as in rounds 2-4 (16,384 blocks and 2 Mbit at least: the host PARALLEL decoder's own line).
// Since round 5 also short ones - at least kDevDecodeMinBlocks blocks (a 256 x 256 frame) and kDevDecodeMinBits stream bits: with two
// launches instead of four the device decoder beats the host's serial decoder far below the old line, the reference's own benchmark
// images (512 x 512, tests/benchmark.py) decompress() in 86 - 100 us instead of 132 - 387.  Round 5 kept streams below 32 bits per block
// on the host (kDevDecodeMinDensity): sparse streams hold blocks longer than the 544-bit range their average asks for, the stitch gave
// up and the second run cost more than the host decoder.  Round 6: the stitch follows the chain over such ranges (tic_entropy_dec_gpu.hip,
// the fix-up loop), every density is taken, and the bit floor is 1 KB (the benchmark set's sparsest stream at q = 5 has 4,071 bytes).
// Hooks TIC_DECODE_MIN_BLOCKS / _BITS / _DENSITY move the lines for measurements.
constexpr size_t kDevDecodeMinBlocks = 1024, kDevDecodeMinBits = 1u << 13, kDevDecodeMinDensity = 0;
static bool device_decoder_takes(size_t n, size_t len) {
    size_t min_blocks = kDevDecodeMinBlocks, min_bits = kDevDecodeMinBits, min_density = kDevDecodeMinDensity;
    if (const char *e = test_hook("TIC_DECODE_MIN_BLOCKS")) min_blocks = (size_t)atol(e);
    if (const char *e = test_hook("TIC_DECODE_MIN_BITS")) min_bits = (size_t)atol(e);
    if (const char *e = test_hook("TIC_DECODE_MIN_DENSITY")) min_density = (size_t)atol(e);
    if (min_bits < 8192) min_bits = 8192; // (entropy_decode_idct_gpu: a stream of at least 128 + 2 x 2,048 bits)
    const size_t bits = len * 8;
    if (bits + 8192 >= (1ull << 32) || bits < 128) return false;
    const bool long_one = n >= 16384 && bits >= 128 + (1u << 21);
    const bool short_one = n >= min_blocks && bits >= 128 + min_bits && bits - 128 >= min_density * n;
    return long_one || short_one;
}

// stream bits per lane of the device decoder: `mult` average blocks, at least `floor_words` 32-bit words, as an odd number of words up to 63
// Round 6's rule: TWO average blocks, at least 288 bits (rounds 3-5: three, at least 544 - then a range in which the walk did not fall in step with the
// chain ended the run; now it hands its exit on, and ranges below 544 bits get eight shadows in front of a wave's own).  tools/range_rule_sweep.py,
// profiles/r06_decoder.txt (7): a 512^2 stream at q = 50 / 10 76 / 71 us instead of 88 / 85, 4096^2 noise at q = 90 / 10 121 / 70 instead of 129 / 76;
// second runs in 600 stress streams 4 instead of 2 (all below 32 bits per block).
static int decode_range_bits(size_t len, size_t n, size_t mult = 2, size_t floor_words = 9) {
    // (nearly flat content - below 7 stream bits per block: DC code + end-of-block and little else - is periodic bit patterns in which a walk can stay
    //  out of step for tens of ranges: ranges twice as long there.  tools/stress_decoder.py 600: second runs on valid streams of 4-7 bits per block 1 in
    //  24 instead of 4; everywhere else the longer range only costs - the benchmark loop's decompress() +10-17 us, profiles/r06_decoder.txt)
    if (floor_words == 9 && len * 8 < 7 * n) floor_words = 33;
    size_t k = (mult * (len * 8) / n + 31) / 32;
    k |= 1;
    return (int)(k < floor_words ? floor_words : (k > 63 ? 63 : k)) * 32;
}

static int decode_on_device(tic_ctx *ctx, const uint8_t *data, size_t len, int h, int w, int quality, int scaled_exp, uint8_t *out,
                            bool out_on_device, size_t out_stride, bool *done, const uint8_t *head16 /* the 16 header bytes h, w, quality came from */,
                            bool src_on_device = false, bool head_is_guess = false, bool *guess_held = nullptr) {
    *done = false;
    if (guess_held) *guess_held = false;
    const uint8_t *guessed_head = head_is_guess ? head16 : nullptr;
    const size_t n = num_blocks(h, w);
    // the host parallel decoder's own threshold: shorter streams are decoded serially in well under a millisecond
    if (!device_decoder_takes(n, len) || test_hook("TIC_DECODE_SERIAL") || test_hook("TIC_DECODE_HOST")) return TIC_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_dec_luts) {
        DecLutsDev *l = new DecLutsDev();
        dec_luts_fill(l->dc11, l->ac11, l->ac16);
        dec_chain_luts_fill(l->mdc, l->mac, l->mlong);
        dec_pair_luts_fill(l->ac2, l->long32); // (new DecLutsDev() zeroed the entries behind the long codewords)
        hipError_t e = hipMalloc((void **)&ctx->d_dec_luts, sizeof(DecLutsDev));
        if (e == hipSuccess) e = hipMemcpy(ctx->d_dec_luts, l, sizeof(DecLutsDev), hipMemcpyHostToDevice);
        delete l;
        if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_dec_status, 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->d_dec_status, ctx->h_dec_status, 0);
        if (e != hipSuccess) return set_err(ctx, TIC_E_HIP, "device decoder set-up failed: %s", hipGetErrorString(e));
    }
    const size_t pitch = align_up((size_t)w, 256);
    int rc = ensure_scratch(ctx, pitch * (size_t)h, n * 128);
    if (rc) return rc;
    // The stream is decoded where it lies when it is in device memory at a 4-byte aligned address (the kernels mask the bytes behind
    // its end themselves); a host stream, or an odd address, goes through the context's stream buffer.
    const bool in_place = src_on_device && ((uintptr_t)data & 3u) == 0;
    if (!in_place) {
        const size_t padded = align_up(len, 4) + 16;
        if (padded > ctx->d_stream_cap) {
            if (ctx->d_stream_buf) HIPCHK(ctx, hipFree(ctx->d_stream_buf));
            ctx->d_stream_buf = nullptr;
            ctx->d_stream_cap = 0;
            HIPCHK(ctx, hipMalloc(&ctx->d_stream_buf, padded));
            ctx->d_stream_cap = padded;
        }
    }
    const void *d_stream = in_place ? (const void *)data : (const void *)ctx->d_stream_buf;
    if (!ctx->h_dec_tail) { // pinned: the stream's last bytes on their way down (device source), the tail's coefficients on their way up
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_dec_tail, kDecTailEnd + kDecTailCoef, hipHostMallocDefault));
    }
    const size_t wb = entropy_decode_gpu_work_bytes(len, n);
    if (wb > ctx->dec_work_bytes) {
        if (ctx->d_dec_work) HIPCHK(ctx, hipFree(ctx->d_dec_work));
        ctx->d_dec_work = nullptr;
        ctx->dec_work_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_dec_work, wb));
        ctx->dec_work_bytes = wb;
    }
    size_t dw = entropy_decode_gpu_desc_words(len, n);
    if (dw > ctx->dec_desc_words) { // the look-back words of the decoder's scans: an array of their own, zero or stamped with a past epoch
        dw = dw < 8192 ? 8192 : 2 * dw;
        if (ctx->d_dec_desc) HIPCHK(ctx, hipFree(ctx->d_dec_desc));
        ctx->d_dec_desc = nullptr;
        ctx->dec_desc_words = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_dec_desc, dw * 8));
        HIPCHK(ctx, hipMemset(ctx->d_dec_desc, 0, dw * 8));
        ctx->dec_desc_words = dw;
        ctx->dec_epoch = 0;
    }
    if (!in_place) HIPCHK(ctx, hipMemcpyAsync(ctx->d_stream_buf, data, len, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    // The blocks that start in the stream's last 2048 bits are decoded on the host (below): with a device source their bytes - the
    // last 256 of the stream, 512 taken - come down NOW, in front of the kernels, instead of in a synchronous copy behind them
    // (only for a run WITH the margin, i.e. the second run on a stream whose end is not what a whole stream's is: see below)
    const size_t end_bytes = len < kDecTailEnd ? len : kDecTailEnd;
    bool tail_prefetched = false;
    // where the pixels go: a device destination whose rows are 8-byte aligned takes them straight from the kernels (row stores are
    // cropped to w); anything else gets them from the context's image buffer with one strided copy at the end
    const bool direct = out_on_device && out_stride % 8 == 0 && (uintptr_t)out % 8 == 0;
    // a small image on its way to host memory: the kernels store its rows (8-byte aligned, back to back) into the context's host-mapped
    // buffer, and one memcpy behind the wait that reads the status takes them to the caller - no copy command, no second wait
    // (a 512 x 512 image: 30 us between the last kernel and the end of the read-back, profiles/r05_decoder.txt)
    const size_t pitch8 = align_up((size_t)w, 8);
    const bool host_pix = !out_on_device && pitch8 * (size_t)h <= kDecHostPixBytes && !test_hook("TIC_DECODE_NO_HOSTPIX");
    if (host_pix) {
        rc = ensure_small(ctx, kSmallHostBytes);
        if (rc) return rc;
    }
    DecIdctArgs ia;
    ia.out = direct ? out : host_pix ? ctx->d_small : (uint8_t *)ctx->d_img;
    ia.h = h;
    ia.w = w;
    ia.stride = direct ? (long)out_stride : host_pix ? (long)pitch8 : (long)pitch;
    ia.bw = (w + 7) / 8;
    ia.aligned8 = 1;
    ia.consts = ctx->d_consts + (scaled_exp >= 0 ? 50 : quality); // codec.py:62: quality = 50 on the scaled branch
    ia.scaled = scaled_exp >= 0;
    ia.pow2 = scaled_exp >= 0 ? ldexp(1.0, scaled_exp) : 1.0;
    memcpy(ia.head, head16, 16); // the header these were derived from (a guess, or the stream's own): the fused kernel writes pixels only under it
    // stream bits per lane (decode_range_bits): 2 average blocks, at least 288 bits, as an odd number of 32-bit words up to 63 (noise at q = 50,
    // 220 bits per block: 480; tiled Lenna, 41, and noise at q = 10, 70: 288; noise at q = 90, 404: 864).  The decoder's kernels are one
    // dependent chain per lane, so their time goes with this number (profiles/r04_decoder.txt: the measure kernel 57 us at 672 bits,
    // 76 at 1,024).  A run in which more ranges in a row than the stitch has rounds stay without a synchronisation point (periodic content),
    // or whose blocks do not follow each other where they are decoded, gives up: one more try with the longest range, then the host decoder
    size_t mult = 2, floor_words = 9;
    if (const char *e = test_hook("TIC_DECODE_RULE")) { // "<average blocks per range>,<least words per range>": measurements of the rule itself
        unsigned a = 0, b = 0;
        if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 1 && a <= 64 && b >= 9 && b <= 63) mult = a, floor_words = b | 1;
    }
    int range_bits = decode_range_bits(len, n, mult, floor_words);
    ctx->last_decode_tries = 0;
    if (const char *e = test_hook("TIC_DECODE_RANGE")) range_bits = atoi(e);
    if (!entropy_decode_gpu_range_ok(range_bits)) return set_err(ctx, TIC_E_ARG, "TIC_DECODE_RANGE=%d: not a range the device decoder takes", range_bits);
    // The first run takes the chain to the stream's END (margin 0): a whole stream then leaves nothing to the host - no second wait, no
    // upload, no extra launch (20 us of 194 for a 4096^2 stream).  If anything at all is flagged on that run the stream is not what a whole
    // one is (cut, damaged, too few blocks for its header) and it is decoded once more the way rounds 2-3 did: blocks that start in the last
    // 2,048 bits go to the host's bit-serial decoder, which reproduces the reference's behaviour at a stream's end.
    int margin_bits = test_hook("TIC_DECODE_MARGIN") ? 2048 : 0;
    int flat_grid = 4096; // (tic_entropy_dec_gpu.hip wave_lookback)
    if (const char *e = test_hook("TIC_DECODE_FLAT_GRID")) flat_grid = atoi(e) < 0 ? 0 : atoi(e);
    DecStatus st;
    for (;;) {
        if (margin_bits && src_on_device && !tail_prefetched) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_dec_tail, (const char *)data + (len - end_bytes), end_bytes, hipMemcpyDeviceToHost, ctx->stream));
            tail_prefetched = true;
        }
        memset(ctx->h_dec_status, 0, sizeof(DecStatus)); // (host-mapped; nothing of an earlier call is in flight: every call ends with a drained stream)
        if (++ctx->dec_epoch >= (1u << 22)) { // (the scans carry 24 bits of 2 x epoch: start over on clean words long before a value could recur)
            HIPCHK(ctx, hipMemsetAsync(ctx->d_dec_desc, 0, ctx->dec_desc_words * 8, ctx->stream));
            ctx->dec_epoch = 1;
        }
        ctx->last_decode_tries++;
        ctx->last_decode_range = range_bits;
        HIPCHK(ctx, entropy_decode_idct_gpu(d_stream, len, n, ctx->d_dec_luts, ctx->d_dec_work, ctx->dec_work_bytes, ctx->d_dec_desc,
                                            ctx->dec_desc_words, ctx->dec_epoch, ia, ctx->d_dec_status, range_bits, margin_bits, ctx->stream, flat_grid));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(&st, ctx->h_dec_status, sizeof st); // (host-mapped: the stream has drained)
        if (guessed_head) { // geometry and quality were a guess (tic_decompress_dev): what this run produced counts only if the stream's header is the guessed one
            if (memcmp(st.head, guessed_head, 16) != 0) return TIC_OK;
            *guess_held = true;
        }
        if (test_hook("TIC_DECODE_TRACE"))
            fprintf(stderr, "device decoder run %d: range %d margin %d -> giveup %d, m %llu of %zu, pos_out %llu of %zu bits\n", ctx->last_decode_tries, range_bits,
                    margin_bits, st.giveup, st.m, n, st.pos_out, len * 8);
        if ((st.giveup & 4) && range_bits < 2016) { // (a range without a synchronisation point breaks the chain: whatever else was flagged follows from it)
            range_bits = 2016; // (one retry, with the longest range: a stream that trips the first choice has blocks far above its average)
            continue;
        }
        if (st.giveup != 0 && margin_bits == 0) {
            margin_bits = 2048;
            continue;
        }
        break;
    }
    ctx->last_decode_giveup = st.giveup | ((st.m == 0 || st.m > n) ? 64 : 0);
    if (ctx->last_decode_giveup != 0) return TIC_OK; // the host decoder takes the whole stream
    if (st.m < n) { // the blocks that start in the stream's last 2048 bits: serial on the host, a few KB uploaded, transformed by idct_kernel's block-range form
        const size_t tail_bytes = (n - (size_t)st.m) * 128;
        std::vector<int16_t> big; // (more tail blocks than the pinned buffer holds: cannot happen for a stream the device decoder accepts - 2048 bits are at most 341 blocks)
        int16_t *tail = reinterpret_cast<int16_t *>(ctx->h_dec_tail + kDecTailEnd);
        if (tail_bytes > kDecTailCoef) {
            big.resize(tail_bytes / 2);
            tail = big.data();
        }
        const size_t off = len - end_bytes; // the piece of the stream that came down in front of the kernels starts here
        if (src_on_device && tail_prefetched && (size_t)st.pos_out / 8 >= off) { // (pos_out >= 8 len - 2048: always inside that piece)
            entropy_decode_tail(ctx->h_dec_tail, end_bytes, h, w, (size_t)st.m, (size_t)st.pos_out - off * 8, st.dc_out, tail);
        } else if (src_on_device) {
            const size_t o2 = (size_t)st.pos_out / 8;
            std::vector<uint8_t> end(len - o2);
            HIPCHK(ctx, hipMemcpy(end.data(), (const char *)data + o2, len - o2, hipMemcpyDeviceToHost));
            entropy_decode_tail(end.data(), len - o2, h, w, (size_t)st.m, (size_t)st.pos_out - o2 * 8, st.dc_out, tail);
        } else {
            entropy_decode_tail(data, len, h, w, (size_t)st.m, (size_t)st.pos_out, st.dc_out, tail);
        }
        HIPCHK(ctx, hipMemcpyAsync((char *)ctx->d_coef + (size_t)st.m * 128, tail, tail_bytes, hipMemcpyHostToDevice, ctx->stream));
        IdctArgs a;
        a.coeffs = (const int16_t *)ctx->d_coef;
        a.out = ia.out;
        a.h = h;
        a.w = w;
        a.stride = ia.stride;
        a.bw = ia.bw;
        a.tiles_x = (a.bw + 7) / 8;
        a.first_block = (long)st.m;
        a.nblocks_sel = (long)(n - (size_t)st.m);
        a.ntiles = (int)((a.nblocks_sel + 7) / 8);
        a.aligned8 = 1;
        a.consts = ia.consts;
        a.scaled = ia.scaled;
        a.pow2 = ia.pow2;
        HIPCHK(ctx, launch_idct(a, ctx->stream));
        if (direct || host_pix || !big.empty()) HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // (the caller's pixels are complete when this returns)
    }
    if (host_pix) { // (the stream has drained: the kernels' stores have arrived)
        if (pitch8 == (size_t)w) memcpy(out, ctx->h_small, (size_t)h * (size_t)w);
        else for (int y = 0; y < h; y++) memcpy(out + (size_t)y * (size_t)w, ctx->h_small + (size_t)y * pitch8, (size_t)w);
    } else if (!direct) {
        HIPCHK(ctx, hipMemcpy2DAsync(out, out_on_device ? out_stride : (size_t)w, ctx->d_img, pitch, (size_t)w, (size_t)h,
                                     out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *done = true;
    return TIC_OK;
}

int tic_idctq(tic_ctx *ctx, const int16_t *coeffs_zz, int h, int w, int quality, uint8_t *out, size_t cap) {
    TIC_LOCK(ctx);
    return idctq_impl(ctx, coeffs_zz, h, w, quality, -1, out, cap);
}

int tic_idctq_scaled(tic_ctx *ctx, const int16_t *coeffs_zz, int h, int w, int exponent, uint8_t *out, size_t cap) {
    TIC_LOCK(ctx);
    if (ctx && exponent < 0) return set_err(ctx, TIC_E_QUALITY, "scaled_dct exponent %d outside 0..62", exponent);
    return idctq_impl(ctx, coeffs_zz, h, w, 50, exponent, out, cap);
}

// Which decoder the last tic_decompress of this context used: 1 = device Huffman decoder, 2 = host decoder (short streams,
// anything unusual in a long one), 0 = none yet.
int tic_last_decode_path(tic_ctx *ctx) {
    TIC_LOCK(ctx);
    return ctx ? ctx->last_decode_path : TIC_E_ARG;
}

// Diagnostics: the device decoder's reason for leaving the last long stream to the host decoder (0: it did not; bits: 1 an incident
// in the first range, 4 a range without a synchronisation point, 8 / 16 / 32 an incident on the true chain, 2 trace overflow, 64 no block produced).
int tic_last_decode_giveup(tic_ctx *ctx) {
    TIC_LOCK(ctx);
    return ctx ? ctx->last_decode_giveup : TIC_E_ARG;
}

// ... and the stream bits per lane its last run worked with, and how many runs the last long stream took (2: the first choice of
// range met a range without a synchronisation point and the longest range was tried).  Either pointer may be null.
int tic_last_decode_range(tic_ctx *ctx, int *range_bits, int *tries) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    if (range_bits) *range_bits = ctx->last_decode_range;
    if (tries) *tries = ctx->last_decode_tries;
    return TIC_OK;
}

int tic_last_decode_guess(tic_ctx *ctx) {
    TIC_LOCK(ctx);
    return ctx ? ctx->last_decode_guess : TIC_E_ARG;
}

int tic_set_decode_guess(tic_ctx *ctx, int enable) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    ctx->dec_guess_on = enable != 0;
    return TIC_OK;
}

int tic_decompress(tic_ctx *ctx, const uint8_t *data, size_t len, uint8_t *out, size_t cap) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    ctx->last_decode_giveup = 0;
    ctx->last_decode_tries = 0;
    int h, w, quality;
    uint32_t flag;
    if (parse_header(data, len, &h, &w, &quality, &flag) != TIC_OK)
        return set_err(ctx, TIC_E_STREAM, "stream shorter than the 16-byte header");
    if (flag & (1u << 31)) return set_err(ctx, TIC_E_STREAM, "streams with an embedded Huffman table are not supported");
    const bool scaled = (flag & (1u << 30)) != 0; // a stream of the reference's C encoder (codec.py:127-128): quality = exponent
    if (h < 0 || w < 0) return set_err(ctx, TIC_E_STREAM, "bad geometry in header");
    if (scaled && (quality < 0 || quality > 62)) return set_err(ctx, TIC_E_QUALITY, "scaled_dct exponent %d in header outside 0..62", quality);
    if (!scaled && (quality < 1 || quality > 99)) return set_err(ctx, TIC_E_QUALITY, "quality %d in header outside 1..99", quality);
    const size_t n = num_blocks(h, w);
    if (n == 0) return TIC_OK;
    if (!out || (size_t)h * (size_t)w > cap) return set_err(ctx, TIC_E_SPACE, "output buffer too small");
    {   // long streams: the Huffman decode runs on the device too; only the stream goes up and the pixels come down
        bool done = false;
        const int rc = decode_on_device(ctx, data, len, h, w, scaled ? 50 : quality, scaled ? quality : -1, out, false, 0, &done, data);
        if (rc) return rc;
        if (done) {
            ctx->last_decode_path = 1;
            return TIC_OK;
        }
    }
    // coefficients land in a pinned buffer kept on the context: no page faults on a fresh 32 MB vector per call, and the upload
    // runs at PCIe speed instead of through the runtime's staging of pageable memory
    if (n * 128 > ctx->h_zz_bytes) {
        if (ctx->h_zz) (void)hipHostFree(ctx->h_zz);
        ctx->h_zz = nullptr;
        ctx->h_zz_bytes = 0;
        HIPCHK(ctx, hipSetDevice(ctx->device));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_zz, n * 128, hipHostMallocDefault));
        ctx->h_zz_bytes = n * 128;
    }
    entropy_decode(data, len, h, w, ctx->h_zz);
    ctx->last_decode_path = 2;
    return idctq_impl(ctx, ctx->h_zz, h, w, scaled ? 50 : quality, scaled ? quality : -1, out, cap);
}

// decompress() of MANY streams at once (the mirror of tic_compress_batch; the reference's benchmark loop, tests/benchmark.py:12-23, decodes
// 49 streams of 512 x 512 per quality one call after the other: 94-105 us each, launch and copy latency).  The streams of a chunk are packed
// into one pinned buffer behind their descriptors and go up in ONE copy; ONE measure launch and ONE fused launch decode all of them (the
// batch forms of the two kernels: every wave and workgroup looks up its frame, nothing crosses a frame - the DC sum starts over with every
// frame, codec.py:53); the pixels of the chunk come down in ONE copy - straight into the caller's buffers where they follow each other in
// memory (pinned for the call by one hipHostRegister), else through the context's pinned buffer and a few copy threads.
// Frame i: what tic_decompress(ctx, streams[i], lens[i], outs[i], caps[i]) gives, geometry in hs[i] / ws[i] (either may be null).  A frame the
// batch kernels do not take (a short or damaged stream, a C-encoder stream, anything the device decoder flags) is decoded by that very
// call afterwards.  Errors: headers are checked before any work (the first bad frame's error, nothing decoded); an error while decoding is
// the first failing frame's, the other frames are complete.
int tic_decompress_batch(tic_ctx *ctx, const uint8_t *const *streams, const size_t *lens, int n, uint8_t *const *outs, const size_t *caps, int *hs, int *ws) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    if (n < 0 || (n > 0 && (!streams || !lens || !outs || !caps))) return set_err(ctx, TIC_E_ARG, "bad batch arguments");
    ctx->last_dbatch_frames = ctx->last_dbatch_fallback = ctx->last_dbatch_chunks = ctx->last_dbatch_direct = 0;
    ctx->bt = BatchTrace(); // (tic_last_batch_phases: [0] packing the upload buffer, [1] enqueue, [2] wait + download, [4] hand-out, [5] single-frame calls)
    if (n == 0) return TIC_OK;
    struct Fr { int h, w, q; size_t nblk; bool batch; };
    std::vector<Fr> fr((size_t)n);
    for (int i = 0; i < n; i++) { // the checks of tic_decompress, for every frame, before any work
        int h, w, quality;
        uint32_t flag;
        if (!streams[i] || parse_header(streams[i], lens[i], &h, &w, &quality, &flag) != TIC_OK) return set_err(ctx, TIC_E_STREAM, "frame %d: stream shorter than the 16-byte header", i);
        if (flag & (1u << 31)) return set_err(ctx, TIC_E_STREAM, "frame %d: streams with an embedded Huffman table are not supported", i);
        const bool scaled = (flag & (1u << 30)) != 0;
        if (h < 0 || w < 0) return set_err(ctx, TIC_E_STREAM, "frame %d: bad geometry in header", i);
        if (scaled && (quality < 0 || quality > 62)) return set_err(ctx, TIC_E_QUALITY, "frame %d: scaled_dct exponent %d in header outside 0..62", i, quality);
        if (!scaled && (quality < 1 || quality > 99)) return set_err(ctx, TIC_E_QUALITY, "frame %d: quality %d in header outside 1..99", i, quality);
        const size_t nb = num_blocks(h, w);
        if (nb && (!outs[i] || (size_t)h * (size_t)w > caps[i])) return set_err(ctx, TIC_E_SPACE, "frame %d: output buffer too small", i);
        if (hs) hs[i] = h;
        if (ws) ws[i] = w;
        fr[(size_t)i] = {h, w, quality, nb, nb != 0 && !scaled && device_decoder_takes(nb, lens[i]) && !test_hook("TIC_DECODE_HOST") && !test_hook("TIC_DECODE_SERIAL")};
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_dec_luts) { // (as decode_on_device: the tables go up with the context's first device decode)
        std::unique_ptr<DecLutsDev> l(new DecLutsDev());
        dec_luts_fill(l->dc11, l->ac11, l->ac16);
        dec_chain_luts_fill(l->mdc, l->mac, l->mlong);
        dec_pair_luts_fill(l->ac2, l->long32);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_dec_luts, sizeof(DecLutsDev)));
        HIPCHK(ctx, hipMemcpy(ctx->d_dec_luts, l.get(), sizeof(DecLutsDev), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_dec_status, 64, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->d_dec_status, ctx->h_dec_status, 0));
    }
    tic_ctx::DecBatch &B = ctx->dbat;
    auto grow_dev = [&](void **p, size_t *cap, size_t need) -> int {
        if (need <= *cap) return TIC_OK;
        if (*p) HIPCHK(ctx, hipFree(*p));
        *p = nullptr, *cap = 0;
        HIPCHK(ctx, hipMalloc(p, need + need / 4));
        *cap = need + need / 4;
        return TIC_OK;
    };
    auto grow_pin = [&](uint8_t **p, size_t *cap, size_t need) -> int {
        if (need <= *cap) return TIC_OK;
        if (*p) HIPCHK(ctx, hipHostFree(*p));
        *p = nullptr, *cap = 0;
        HIPCHK(ctx, hipHostMalloc((void **)p, need + need / 4, hipHostMallocDefault));
        *cap = need + need / 4;
        return TIC_OK;
    };
    // chunks: frames in order, while the chunk's streams, pixels and frame count stay inside the limits (one 512 x 512 benchmark set - 49
    // frames, 12.8 MB of pixels - is one chunk; sixteen 4096^2 frames are one)
    constexpr size_t kMaxIn = 96u << 20, kMaxPix = 288u << 20;
    constexpr int kMaxFrames = 1024;
    int result = TIC_OK;
    auto fail = [&](int rc) { if (result == TIC_OK) result = rc; };
    std::vector<int> later; // frames for the single-frame call
    for (int i = 0; i < n; i++)
        if (!fr[(size_t)i].batch && fr[(size_t)i].nblk) later.push_back(i);
    int i0 = 0;
    while (i0 < n) {
        // ---- the chunk's frames and the layout of its buffers
        std::vector<int> ids;
        size_t in_bytes = 0, pix_bytes = 0, ranges288 = 0, blocks = 0;
        int range_bits = 0;
        bool small_win = true;
        int i1 = i0;
        for (; i1 < n; i1++) {
            const Fr &f = fr[(size_t)i1];
            if (!f.batch) continue;
            const size_t pitch = (size_t)(f.w % 8 == 0 ? f.w : (f.w + 7) / 8 * 8);
            const size_t sb = align_up(lens[i1], 16) + 16, pb = align_up(pitch * (size_t)f.h, 256);
            if (!ids.empty() && (in_bytes + sb > kMaxIn || pix_bytes + pb > kMaxPix || (int)ids.size() >= kMaxFrames)) break;
            ids.push_back(i1);
            in_bytes += sb, pix_bytes += pb, blocks += f.nblk, ranges288 += lens[i1] * 8 / 288 + 2;
            const int rb = decode_range_bits(lens[i1], f.nblk);
            range_bits = rb > range_bits ? rb : range_bits;
            small_win = small_win && lens[i1] * 8 / f.nblk <= 240;
        }
        i0 = i1;
        if (ids.empty()) break;
        const uint32_t F = (uint32_t)ids.size();
        std::vector<DecFrame> frames(F);
        std::vector<size_t> pix_off(F), pitches(F);
        uint32_t tiles = 0, wgs = 0, ranges = 0;
        size_t blk = 0, words = 0, poff = 0;
        for (uint32_t k = 0; k < F; k++) {
            const int i = ids[k];
            const Fr &f = fr[(size_t)i];
            DecFrame &d = frames[k];
            const size_t len = lens[i];
            d.word0 = (uint32_t)words;
            d.nwords = (uint32_t)((len + 3) / 4);
            d.last_mask = (len & 3) ? 0xffffffffu << (8u * (4u - (uint32_t)(len & 3))) : 0xffffffffu;
            d.stream_bits = d.fast_end = (uint32_t)(len * 8);
            d.nranges = entropy_decode_batch_ranges(len, range_bits);
            d.range0 = ranges;
            d.tile0 = tiles;
            d.ntiles = entropy_decode_batch_tiles(d.nranges, range_bits);
            d.blk0 = (uint32_t)blk;
            d.nblocks = (uint32_t)f.nblk;
            d.wg0 = wgs;
            d.nwgs = entropy_decode_batch_wgs(f.nblk);
            d.pad_ = 0;
            pitches[k] = (size_t)(f.w % 8 == 0 ? f.w : (f.w + 7) / 8 * 8);
            pix_off[k] = poff;
            d.idct.out = nullptr; // (set below, when the pixel buffer exists)
            d.idct.h = f.h, d.idct.w = f.w;
            d.idct.stride = (long)pitches[k];
            d.idct.bw = (f.w + 7) / 8;
            d.idct.aligned8 = 1;
            d.idct.consts = ctx->d_consts + f.q;
            d.idct.scaled = 0;
            d.idct.pow2 = 1.0;
            memcpy(d.idct.head, streams[i], 16);
            words += (align_up(len, 16) + 16) / 4;
            ranges += d.nranges, tiles += d.ntiles, wgs += d.nwgs, blk += f.nblk;
            poff += align_up(pitches[k] * (size_t)f.h, 256);
        }
        // one upload: descriptors, the frame of every measure wave, the frame of every fused workgroup, the streams
        const size_t o_frames = 0, o_tiles = align_up(o_frames + F * sizeof(DecFrame), 256), o_wgs = align_up(o_tiles + (size_t)tiles * 4, 256),
                     o_streams = align_up(o_wgs + (size_t)wgs * 4, 256), up_bytes = o_streams + words * 4;
        if (up_bytes > B.in_cap) { // the pinned upload buffer and its device mirror grow together
            if (B.h_in) HIPCHK(ctx, hipHostFree(B.h_in));
            if (B.d_in) HIPCHK(ctx, hipFree(B.d_in));
            B.h_in = B.d_in = nullptr, B.in_cap = 0;
            const size_t cap = up_bytes + up_bytes / 4;
            HIPCHK(ctx, hipHostMalloc((void **)&B.h_in, cap, hipHostMallocDefault));
            HIPCHK(ctx, hipMalloc((void **)&B.d_in, cap));
            B.in_cap = cap;
        }
        int rc = TIC_OK;
        rc = grow_dev((void **)&B.d_pix, &B.pix_cap, poff);
        if (rc) return rc;
        rc = grow_dev(&B.d_work, &B.work_bytes, entropy_decode_batch_work_bytes(ranges288, blocks, F));
        if (rc) return rc;
        {
            size_t dw = 4 * (size_t)((tiles > wgs ? tiles : wgs) + 2);
            if (dw > B.desc_words) {
                dw = dw < 8192 ? 8192 : 2 * dw;
                if (B.d_desc) HIPCHK(ctx, hipFree(B.d_desc));
                B.d_desc = nullptr, B.desc_words = 0;
                HIPCHK(ctx, hipMalloc((void **)&B.d_desc, dw * 8));
                HIPCHK(ctx, hipMemset(B.d_desc, 0, dw * 8));
                B.desc_words = dw;
                B.epoch = 0;
            }
            if (F > B.status_cap) {
                if (B.h_status) HIPCHK(ctx, hipHostFree(B.h_status));
                B.h_status = nullptr, B.status_cap = 0;
                const size_t cap = F < 256 ? 256 : 2 * (size_t)F;
                HIPCHK(ctx, hipHostMalloc((void **)&B.h_status, cap * sizeof(DecStatus), hipHostMallocMapped | hipHostMallocCoherent));
                HIPCHK(ctx, hipHostGetDevicePointer((void **)&B.d_status, B.h_status, 0));
                B.status_cap = cap;
            }
        }
        BT_START();
        for (uint32_t k = 0; k < F; k++) frames[k].idct.out = B.d_pix + pix_off[k];
        memcpy(B.h_in + o_frames, frames.data(), F * sizeof(DecFrame));
        {
            uint32_t *tf = (uint32_t *)(B.h_in + o_tiles), *wf = (uint32_t *)(B.h_in + o_wgs);
            for (uint32_t k = 0; k < F; k++) {
                for (uint32_t t = 0; t < frames[k].ntiles; t++) tf[frames[k].tile0 + t] = k;
                for (uint32_t g = 0; g < frames[k].nwgs; g++) wf[frames[k].wg0 + g] = k;
            }
            for (uint32_t k = 0; k < F; k++) { // (the bytes behind a stream's last word are never read: no need to clear them)
                memcpy(B.h_in + o_streams + (size_t)frames[k].word0 * 4, streams[ids[k]], lens[ids[k]]);
            }
        }
        memset(B.h_status, 0, F * sizeof(DecStatus));
        BT_STOP(0);
        BT_START();
        if (++B.epoch >= (1u << 22)) {
            HIPCHK(ctx, hipMemsetAsync(B.d_desc, 0, B.desc_words * 8, ctx->stream));
            B.epoch = 1;
        }
        HIPCHK(ctx, hipMemcpyAsync(B.d_in, B.h_in, up_bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, entropy_decode_idct_gpu_batch(B.d_in + o_streams, (const DecFrame *)(B.d_in + o_frames), (const uint32_t *)(B.d_in + o_tiles), (const uint32_t *)(B.d_in + o_wgs), F, tiles,
                                                  wgs, ranges, blk, small_win, ctx->d_dec_luts, B.d_work, B.work_bytes, B.d_desc, B.desc_words, B.epoch, B.d_status, range_bits,
                                                  ctx->stream));
        BT_STOP(1);
        BT_START();
        // ---- the pixels come down: one copy into the caller's memory where the frames are dense and follow each other there, else one copy
        // into pinned memory and a few threads
        bool dense = true;
        for (uint32_t k = 0; k < F && dense; k++)
            dense = pitches[k] == (size_t)fr[(size_t)ids[k]].w && (k == 0 || (outs[ids[k]] == outs[ids[k - 1]] + (pix_off[k] - pix_off[k - 1])));
        bool direct = false;
        if (dense && ctx->auto_register) {
            const size_t total = pix_off[F - 1] + (size_t)fr[(size_t)ids[F - 1]].h * (size_t)fr[(size_t)ids[F - 1]].w;
            bool pinned = host_pointer_is_pinned(outs[ids[0]]) && host_pointer_is_pinned(outs[ids[0]] + total - 1);
            void *reg = nullptr;
            if (!pinned && total >= (256u << 10)) {
                const uintptr_t lo = (uintptr_t)outs[ids[0]] & ~(uintptr_t)4095, hi = ((uintptr_t)outs[ids[0]] + total + 4095) & ~(uintptr_t)4095;
                if (hipHostRegister((void *)lo, hi - lo, hipHostRegisterDefault) == hipSuccess) {
                    reg = (void *)lo;
                    pinned = true;
                } else
                    (void)hipGetLastError();
            }
            if (pinned) {
                hipError_t e = hipMemcpyAsync(outs[ids[0]], B.d_pix, total, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                direct = e == hipSuccess;
                if (!direct) (void)hipGetLastError();
            }
            if (reg && hipHostUnregister(reg) != hipSuccess) (void)hipGetLastError();
        }
        if (!direct) {
            rc = grow_pin(&B.h_pix, &B.hpix_cap, poff);
            if (rc) return rc;
            HIPCHK(ctx, hipMemcpyAsync(B.h_pix, B.d_pix, poff, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        BT_STOP(2);
        ctx->last_dbatch_chunks++;
        ctx->last_dbatch_direct += direct ? (int)F : 0;
        // ---- what the kernels report, frame by frame: the header they saw, nothing flagged, every block produced
        std::vector<char> good(F);
        for (uint32_t k = 0; k < F; k++) {
            const DecStatus &st = B.h_status[k];
            good[k] = memcmp(st.head, streams[ids[k]], 16) == 0 && st.giveup == 0 && st.m == (unsigned long long)frames[k].nblocks;
            if (good[k]) ctx->last_dbatch_frames++;
            else later.push_back(ids[k]);
        }
        BT_START();
        if (!direct) {
            auto hand_out = [&](int t, int T) {
                if (T > 1) bind_pipeline_thread(ctx);
                for (uint32_t k = (uint32_t)t; k < F; k += (uint32_t)T) {
                    if (!good[k]) continue;
                    const Fr &f = fr[(size_t)ids[k]];
                    const uint8_t *src = B.h_pix + pix_off[k];
                    if (pitches[k] == (size_t)f.w) memcpy(outs[ids[k]], src, (size_t)f.h * (size_t)f.w);
                    else for (int y = 0; y < f.h; y++) memcpy(outs[ids[k]] + (size_t)y * (size_t)f.w, src + (size_t)y * pitches[k], (size_t)f.w);
                }
            };
            const int T = poff < (2u << 20) || F < 2 ? 1 : (F < 8 ? (int)F : 8);
            if (T == 1) hand_out(0, 1);
            else {
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back(hand_out, t, T);
                for (auto &x : th) x.join();
            }
        }
        BT_STOP(4);
    }
    // frames the batch did not take, or did not finish: the single-frame call, with everything it knows (second run, host decoders)
    std::sort(later.begin(), later.end());
    for (int i : later) {
        const int rc = tic_decompress(ctx, streams[i], lens[i], outs[i], caps[i]);
        ctx->last_dbatch_fallback++;
        if (rc != TIC_OK) {
            if (result == TIC_OK) {
                const std::string m = ctx->err;
                set_err(ctx, rc, "frame %d: %s", i, m.c_str());
            }
            fail(rc);
        }
    }
    return result;
}

// How the last tic_decompress_batch went: frames decoded by the batch kernels, frames that took the single-frame call (too short for the
// device decoder, C-encoder streams, anything flagged), chunks, frames whose pixels were copied straight into the caller's memory.
int tic_last_decompress_batch(tic_ctx *ctx, int *batch_frames, int *single_frames, int *chunks, int *direct_frames) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    if (batch_frames) *batch_frames = ctx->last_dbatch_frames;
    if (single_frames) *single_frames = ctx->last_dbatch_fallback;
    if (chunks) *chunks = ctx->last_dbatch_chunks;
    if (direct_frames) *direct_frames = ctx->last_dbatch_direct;
    return TIC_OK;
}

// decompress() with stream and pixels both resident in HBM (the counterpart of tic_compress_dev): only the 16-byte header, the
// status and - for the blocks that start in the stream's last 2048 bits - a few hundred bytes cross PCIe.  Short streams and streams
// the device decoder gives up on come down to the host decoder and their coefficients go back up (the reference's behaviour on
// malformed streams lives there).  d_out: h rows of w pixels, out_stride bytes apart.
int tic_decompress_dev(tic_ctx *ctx, const void *d_stream, size_t len, void *d_out, ptrdiff_t out_stride, size_t out_cap, int *h_out, int *w_out) {
    TIC_LOCK(ctx);
    if (!ctx) return TIC_E_ARG;
    ctx->last_decode_giveup = 0;
    ctx->last_decode_tries = 0;
    ctx->last_decode_guess = 0;
    if (!d_stream && len) return set_err(ctx, TIC_E_ARG, "null stream pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (len < 16) return set_err(ctx, TIC_E_STREAM, "stream shorter than the 16-byte header");
    // Geometry and quality are in the stream's header - in device memory; a synchronous 16-byte read in front of the first launch is a
    // sixth of the call for a 4096^2 stream.  A long stream is therefore decoded on a GUESS - the header of the stream this context decoded
    // last (frames of a sequence, the images of a batch) - and the first kernel echoes the real header into the status words: when it is
    // the guessed one, everything the run produced stands, without the read; when it is not (or the guess does not fit this call's
    // buffers), the header is read and the stream decoded again.  A run on a wrong guess writes NO pixel: the fused kernel compares
    // the stream's first 16 bytes with the header its geometry came from before it stores anything (DecIdctArgs::head), so nothing
    // outside the real h x w is ever touched - a destination that is a window of a larger surface keeps its neighbours
    // (test_decompress_dev_wrong_guess_writes_nothing_outside_the_image).  (On an error return the contents of d_out are unspecified.)
    // ... and only after two streams in a row came with the same header (dec_head_streak): alternating geometries never pay for a guess,
    // a change behind a run of equal frames pays once.  tic_set_decode_guess(ctx, 0) turns the guessing off.
    const bool may_guess = ctx->dec_guess_on && ctx->dec_head_valid && ctx->dec_head_streak >= 1 && device_decoder_takes(kDevDecodeMinBlocks, len) &&
                           !test_hook("TIC_DECODE_NO_GUESS");
    for (int attempt = may_guess ? 0 : 1; attempt < 2; attempt++) {
        const bool guess = attempt == 0;
        uint8_t head[16] = {0};
        if (guess)
            memcpy(head, ctx->dec_head, 16);
        else
            HIPCHK(ctx, hipMemcpy(head, d_stream, 16, hipMemcpyDeviceToHost));
        int h, w, quality;
        uint32_t flag;
        if (parse_header(head, 16, &h, &w, &quality, &flag) != TIC_OK) return set_err(ctx, TIC_E_STREAM, "stream shorter than the 16-byte header");
        if (flag & (1u << 31)) return set_err(ctx, TIC_E_STREAM, "streams with an embedded Huffman table are not supported"); // (a cached header passed these checks when it was cached)
        const bool scaled = (flag & (1u << 30)) != 0;
        if (h < 0 || w < 0) return set_err(ctx, TIC_E_STREAM, "bad geometry in header");
        if (scaled && (quality < 0 || quality > 62)) return set_err(ctx, TIC_E_QUALITY, "scaled_dct exponent %d in header outside 0..62", quality);
        if (!scaled && (quality < 1 || quality > 99)) return set_err(ctx, TIC_E_QUALITY, "quality %d in header outside 1..99", quality);
        const size_t n = num_blocks(h, w);
        if (guess && (n == 0 || out_stride < (ptrdiff_t)w || !d_out || (size_t)(h - 1) * (size_t)out_stride + (size_t)w > out_cap)) { // the guess does not fit this call
            ctx->last_decode_guess = -1;
            continue;
        }
        if (!guess) {
            if (h_out) *h_out = h;
            if (w_out) *w_out = w;
            if (n == 0) return TIC_OK;
            if (out_stride < (ptrdiff_t)w) return set_err(ctx, TIC_E_ARG, "row stride %td smaller than the width %d", out_stride, w);
            if (!d_out || (size_t)(h - 1) * (size_t)out_stride + (size_t)w > out_cap) return set_err(ctx, TIC_E_SPACE, "output buffer too small");
        }
        bool done = false, held = false;
        int rc = decode_on_device(ctx, (const uint8_t *)d_stream, len, h, w, scaled ? 50 : quality, scaled ? quality : -1, (uint8_t *)d_out, true,
                                  (size_t)out_stride, &done, head, true, guess, guess ? &held : nullptr);
        if (rc) return rc;
        if (guess) {
            ctx->last_decode_guess = held ? 1 : -1;
            if (!held) continue; // another header (or the device decoder does not take this stream): read it
            if (h_out) *h_out = h;
            if (w_out) *w_out = w;
        }
        if (done) {
            ctx->last_decode_path = 1;
            ctx->dec_head_streak = ctx->dec_head_valid && memcmp(ctx->dec_head, head, 16) == 0 ? (ctx->dec_head_streak < 1000 ? ctx->dec_head_streak + 1 : 1000) : 0;
            memcpy(ctx->dec_head, head, 16);
            ctx->dec_head_valid = true;
            return TIC_OK;
        }
        std::vector<uint8_t> host(len);
        HIPCHK(ctx, hipMemcpy(host.data(), d_stream, len, hipMemcpyDeviceToHost));
        if (n * 128 > ctx->h_zz_bytes) {
            if (ctx->h_zz) (void)hipHostFree(ctx->h_zz);
            ctx->h_zz = nullptr;
            ctx->h_zz_bytes = 0;
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_zz, n * 128, hipHostMallocDefault));
            ctx->h_zz_bytes = n * 128;
        }
        entropy_decode(host.data(), len, h, w, ctx->h_zz);
        ctx->last_decode_path = 2;
        rc = ensure_scratch(ctx, align_up((size_t)w, 256) * (size_t)h, n * 128);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_coef, ctx->h_zz, n * 128, hipMemcpyHostToDevice, ctx->stream));
        return idct_from_device(ctx, h, w, scaled ? 50 : quality, scaled ? quality : -1, (uint8_t *)d_out, true, (size_t)out_stride);
    }
    return set_err(ctx, TIC_E_ARG, "tic_decompress_dev: unreachable");
}

// tic_decompress_dev, asynchronously.  A long stream is LAUNCHED on the guess of its header (tic_decompress_dev above) on a stream of
// the ticket's own and the call returns; tic_decompress_async_result waits for it and checks what the kernels reported - the header they
// saw, nothing unusual on the chain, every block produced; if any of that fails (another header, a damaged or cut stream) the stream is
// decoded again there, synchronously, by tic_decompress_dev itself, so the outcome of a ticket is always that of the synchronous call.
// A call that cannot be launched on a guess (no stream decoded yet on this context, a short stream, a destination the kernels cannot
// write directly or the guessed geometry does not fit) runs synchronously right away and its ticket only carries the outcome.
int tic_decompress_dev_async(tic_ctx *ctx, const void *d_stream, size_t len, void *d_out, ptrdiff_t out_stride, size_t out_cap, long long *ticket) {
    TIC_LOCK(ctx);
    if (!ctx || !ticket) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long long t = ctx->dec_async_next;
    tic_ctx::DecSlot &sl = ctx->dec_slots[t % kDecSlots];
    if (sl.ticket >= 0) return set_err(ctx, TIC_E_ARG, "%d asynchronous decodes are open: collect results (tic_decompress_async_result) first", kDecSlots);
    sl.launched = false;
    sl.d_stream = d_stream;
    sl.len = len;
    sl.d_out = d_out;
    sl.out_stride = out_stride;
    sl.out_cap = out_cap;
    int h = 0, w = 0, quality = 0;
    uint32_t flag = 0;
    bool launch = ctx->dec_guess_on && ctx->dec_head_valid && ctx->dec_head_streak >= 1 && d_stream && d_out && ((uintptr_t)d_stream & 3u) == 0 &&
                  out_stride % 8 == 0 && (uintptr_t)d_out % 8 == 0 && ctx->d_dec_luts && !test_hook("TIC_DECODE_NO_GUESS") && !test_hook("TIC_DECODE_HOST") &&
                  !test_hook("TIC_DECODE_SERIAL") && parse_header(ctx->dec_head, 16, &h, &w, &quality, &flag) == TIC_OK;
    const size_t n = launch ? num_blocks(h, w) : 0;
    launch = launch && device_decoder_takes(n, len) && out_stride >= (ptrdiff_t)w && (size_t)(h - 1) * (size_t)out_stride + (size_t)w <= out_cap;
    if (launch) {
        if (!sl.stream) HIPCHK(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        if (!ctx->dec_order) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->dec_order, hipEventDisableTiming));
        if (!sl.h_status) {
            HIPCHK(ctx, hipHostMalloc((void **)&sl.h_status, 64, hipHostMallocMapped | hipHostMallocCoherent));
            HIPCHK(ctx, hipHostGetDevicePointer((void **)&sl.d_status, sl.h_status, 0));
        }
        const size_t wb = entropy_decode_gpu_work_bytes(len, n);
        if (wb > sl.work_bytes) { // (nothing of this slot is in flight: its ticket is closed)
            if (sl.work) HIPCHK(ctx, hipFree(sl.work));
            sl.work = nullptr;
            sl.work_bytes = 0;
            HIPCHK(ctx, hipMalloc(&sl.work, wb));
            sl.work_bytes = wb;
        }
        size_t dw = entropy_decode_gpu_desc_words(len, n);
        if (dw > sl.desc_words) {
            dw = dw < 8192 ? 8192 : 2 * dw;
            if (sl.desc) HIPCHK(ctx, hipFree(sl.desc));
            sl.desc = nullptr;
            sl.desc_words = 0;
            HIPCHK(ctx, hipMalloc((void **)&sl.desc, dw * 8));
            HIPCHK(ctx, hipMemset(sl.desc, 0, dw * 8));
            sl.desc_words = dw;
            sl.epoch = 0;
        }
        if (++sl.epoch >= (1u << 22)) {
            HIPCHK(ctx, hipMemsetAsync(sl.desc, 0, sl.desc_words * 8, sl.stream));
            sl.epoch = 1;
        }
        const bool scaled = (flag & (1u << 30)) != 0;
        DecIdctArgs ia;
        ia.out = (uint8_t *)d_out;
        ia.h = h;
        ia.w = w;
        ia.stride = (long)out_stride;
        ia.bw = (w + 7) / 8;
        ia.aligned8 = 1;
        ia.consts = ctx->d_consts + (scaled ? 50 : quality);
        ia.scaled = scaled;
        ia.pow2 = scaled ? ldexp(1.0, quality) : 1.0;
        memcpy(ia.head, ctx->dec_head, 16); // (the guess: pixels are written only if the stream really starts with it)
        memset(sl.h_status, 0, sizeof(DecStatus));
        // behind everything queued on the context's stream so far (the stream may just have been written there: tic_compress_dev_async)
        HIPCHK(ctx, hipEventRecord(ctx->dec_order, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(sl.stream, ctx->dec_order, 0));
        HIPCHK(ctx, entropy_decode_idct_gpu(d_stream, len, n, ctx->d_dec_luts, sl.work, sl.work_bytes, sl.desc, sl.desc_words, sl.epoch, ia, sl.d_status,
                                            decode_range_bits(len, n), 0, sl.stream));
        HIPCHK(ctx, hipEventRecord(sl.done, sl.stream));
        memcpy(sl.head, ctx->dec_head, 16);
        sl.n = n;
        sl.h = h;
        sl.w = w;
        sl.launched = true;
    } else {
        sl.rc = tic_decompress_dev(ctx, d_stream, len, d_out, out_stride, out_cap, &sl.h, &sl.w);
    }
    sl.ticket = t;
    ctx->dec_async_next = t + 1;
    *ticket = t;
    return TIC_OK;
}

// Outcome of an asynchronous decode: what tic_decompress_dev would have returned for the same arguments (and *h_out / *w_out, either
// may be null).  wait == 0: TIC_E_BUSY while the frame is still in flight.  A ticket is closed by the call that returns anything else.
int tic_decompress_async_result(tic_ctx *ctx, long long ticket, int wait, int *h_out, int *w_out) {
    TIC_LOCK(ctx);
    if (!ctx || ticket < 0) return TIC_E_ARG;
    tic_ctx::DecSlot &sl = ctx->dec_slots[ticket % kDecSlots];
    if (sl.ticket != ticket) return set_err(ctx, TIC_E_ARG, "decode ticket %lld is not open", ticket);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (sl.launched) {
        {   // (as tic_async_result: a failing event call closes the ticket too)
            const hipError_t q = wait ? hipEventSynchronize(sl.done) : hipEventQuery(sl.done);
            if (!wait && q == hipErrorNotReady) return TIC_E_BUSY;
            if (q != hipSuccess) sl.ticket = -1;
            HIPCHK(ctx, q);
        }
        DecStatus st;
        memcpy(&st, sl.h_status, sizeof st); // (host-mapped: the slot's stream has drained)
        if (memcmp(st.head, sl.head, 16) == 0 && st.giveup == 0 && st.m == sl.n) {
            ctx->last_decode_path = 1;
            ctx->last_decode_giveup = 0;
            ctx->last_decode_guess = 1;
            ctx->last_decode_tries = 1;
            sl.rc = TIC_OK;
        } else { // another header, or something unusual in the stream: the synchronous call settles it (and overwrites what this run wrote)
            const bool on = ctx->dec_guess_on;
            ctx->dec_guess_on = false; // (the launch on the guess is what just failed: this time the header is read first)
            sl.rc = tic_decompress_dev(ctx, sl.d_stream, sl.len, sl.d_out, sl.out_stride, sl.out_cap, &sl.h, &sl.w);
            ctx->dec_guess_on = on;
            ctx->last_decode_guess = -1;
        }
    }
    sl.ticket = -1;
    if (h_out) *h_out = sl.h;
    if (w_out) *w_out = sl.w;
    return sl.rc;
}

// ---- self test hook (used by tests/ only; not part of the drop-in surface) --------------------------------
int tic_selftest_transpose(tic_ctx *ctx, const void *host_in, void *host_dpp, void *host_ref, int nthreads) {
    TIC_LOCK(ctx);
    if (!ctx || nthreads <= 0 || nthreads % 256) return TIC_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void *d_in, *d_a, *d_b;
    size_t bytes = (size_t)nthreads * 8;
    HIPCHK(ctx, hipMalloc(&d_in, bytes));
    HIPCHK(ctx, hipMalloc(&d_a, bytes));
    HIPCHK(ctx, hipMalloc(&d_b, bytes));
    HIPCHK(ctx, hipMemcpy(d_in, host_in, bytes, hipMemcpyHostToDevice));
    HIPCHK(ctx, launch_selftest_transpose(d_in, d_a, d_b, nthreads, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(host_dpp, d_a, bytes, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(host_ref, d_b, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_a);
    (void)hipFree(d_b);
    return TIC_OK;
}

} // extern "C"
