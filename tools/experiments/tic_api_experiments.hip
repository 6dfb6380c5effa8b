// The C-ABI translation unit of the EXPERIMENT library (tools/bin/libtinyimgcodec_hip_ablate.so): the product's tic_api.hip,
// unchanged, plus the debug entry points that only the scripts under tools/ call.  They live here - behind the product source,
// inside the same translation unit, so that they see its context structure and helpers - and not in csrc/.
#include "../../tinyimgcodec_amd/csrc/tic_api.hip"

namespace {
void *g_dbg = nullptr; // diagnostic stamp buffer (one per process: the stamp scripts use a single context)
}

extern "C" {
// One launch of a stamp build (variant 17, 52 or 71 of tic_kernels_experiments.hip): per-wave s_memtime stamps to host_out.
int tic_debug_stamps(tic_ctx *ctx, const void *d_image, int h, int w, ptrdiff_t row_stride, int quality, void *d_coeffs_zz,
                     unsigned long long *host_out, size_t n_u64, int variant) {
    TIC_LOCK(ctx);
    int rc = check_geometry(ctx, h, w, row_stride, quality);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t cap = 8192 * 4 * 8;
    if (!g_dbg) HIPCHK(ctx, hipMalloc(&g_dbg, cap * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMemsetAsync(g_dbg, 0, cap * sizeof(unsigned long long), ctx->stream));
    DctqArgs a = make_args(ctx, d_image, h, w, row_stride, quality, d_coeffs_zz);
    a.dbg = (unsigned long long *)g_dbg;
    HIPCHK(ctx, launch_dctq(a, (variant == 52 || variant == 71) ? variant : 17, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n = n_u64 < cap ? n_u64 : cap;
    HIPCHK(ctx, hipMemcpy(host_out, g_dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return TIC_OK;
}
}
