"""What the link gives: pinned host <-> device copies, each direction alone and both at once (two streams), in chunk sizes the batch
pipeline uses (33 MB up, 14 MB down per chunk of 16 1080p frames) and as the strided 2-D read-back it issues (hipMemcpy2DAsync)."""
import ctypes as C, sys, time
hip = C.CDLL("libamdhip64.so")
def chk(r):
    assert r == 0, r
H2D, D2H = 1, 2
def host_alloc(n):
    p = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(p), C.c_size_t(n), 0)); return p
def dev_alloc(n):
    p = C.c_void_p(); chk(hip.hipMalloc(C.byref(p), C.c_size_t(n))); return p
def stream():
    s = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(s), 1)); return s
chk(hip.hipSetDevice(0))
up, down = 16 * 1080 * 1920, 16 * 890880
hu, hd, du, dd = host_alloc(up), host_alloc(16 * 1113600), dev_alloc(up), dev_alloc(16 * 1113600)
s1, s2 = stream(), stream()
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
def run(n, do_up, do_down, two_d=False):
    t = time.perf_counter()
    for _ in range(n):
        if do_up: chk(hip.hipMemcpyAsync(du, hu, up, H2D, s1))
        if do_down:
            if two_d: chk(hip.hipMemcpy2DAsync(hd, 890880, dd, 1113600, 890880, 16, D2H, s2))
            else: chk(hip.hipMemcpyAsync(hd, dd, down, D2H, s2))
    chk(hip.hipStreamSynchronize(s1)); chk(hip.hipStreamSynchronize(s2))
    return (time.perf_counter() - t) / n
for name, a in (("up alone (33 MB)", (True, False)), ("down alone, linear (14 MB)", (False, True)), ("down alone, 2-D strided (16 x 0.89 MB)", (False, True, True)),
                ("both, down linear", (True, True)), ("both, down 2-D strided", (True, True, True))):
    run(3, *a)
    dt = run(32, *a)
    gb = ((up if a[0] else 0) + (down if a[1] else 0)) / dt / 1e9
    print("%-42s %.3f ms per chunk  %.1f GB/s over the link%s" % (name, dt * 1e3, gb, "   -> 16 chunks %.1f ms" % (dt * 16e3) if a[0] and a[1] else ""))

# the pipeline's shape: uploads alternate between TWO streams (with a kernel's worth of work behind each), read-backs on a third
s3 = stream()
def run3(n, down_stream_is_upload_stream=False):
    t = time.perf_counter()
    for k in range(n):
        su = s1 if (k & 1) == 0 else s3
        chk(hip.hipMemcpyAsync(du, hu, up, H2D, su))
        chk(hip.hipMemcpy2DAsync(hd, 890880, dd, 1113600, 890880, 16, D2H, su if down_stream_is_upload_stream else s2))
    for s in (s1, s2, s3): chk(hip.hipStreamSynchronize(s))
    return (time.perf_counter() - t) / n
for name, flag in (("uploads on two streams in turn, read-back on a third", False), ("read-back on the stream of the chunk's upload", True)):
    run3(4, flag)
    dt = run3(32, flag)
    print("%-58s %.3f ms per chunk -> 16 chunks %.1f ms" % (name, dt * 1e3, dt * 16e3))

# does it matter what kind of pinned memory the frames lie in?  hipHostMalloc'ed against ordinary (numpy / malloc) memory pinned with hipHostRegister
import numpy as np
arr = np.random.default_rng(1).integers(0, 256, up, dtype=np.uint8)
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
chk(hip.hipHostRegister(arr.ctypes.data, arr.nbytes, 0))
def run_reg(n):
    t = time.perf_counter()
    for _ in range(n): chk(hip.hipMemcpyAsync(du, arr.ctypes.data, up, H2D, s1))
    chk(hip.hipStreamSynchronize(s1))
    return (time.perf_counter() - t) / n
run_reg(3); dt = run_reg(32)
print("up alone from hipHostRegister'ed memory (33 MB)          %.3f ms per chunk  %.1f GB/s" % (dt * 1e3, up / dt / 1e9))
chk(hip.hipHostUnregister(C.c_void_p(arr.ctypes.data)))
for flags, nm in ((1, "portable"), (2, "mapped"), (3, "portable|mapped")):
    chk(hip.hipHostRegister(arr.ctypes.data, arr.nbytes, flags))
    run_reg(3); dt = run_reg(16)
    print("   ... registered with flags %d (%s): %.3f ms per chunk  %.1f GB/s" % (flags, nm, dt * 1e3, up / dt / 1e9))
    chk(hip.hipHostUnregister(C.c_void_p(arr.ctypes.data)))
# one large registered block, copies from successive 33 MB pieces of it (what the batch pipeline does with a caller-registered batch)
big = np.zeros(16 * up, dtype=np.uint8); big[::4096] = 1
chk(hip.hipHostRegister(big.ctypes.data, big.nbytes, 0))
def run_big(n):
    t = time.perf_counter()
    for k in range(n): chk(hip.hipMemcpyAsync(du, big.ctypes.data + (k % 16) * up, up, H2D, s1))
    chk(hip.hipStreamSynchronize(s1))
    return (time.perf_counter() - t) / n
run_big(16); dt = run_big(32)
print("up alone from successive pieces of one registered 531 MB block: %.3f ms per chunk  %.1f GB/s" % (dt * 1e3, up / dt / 1e9))
