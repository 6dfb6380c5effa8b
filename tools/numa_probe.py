"""Where do the caller's frames live, where does the device sit, and what does it cost?  256 x 1080p host -> host through
tic_compress_batch with the frames first-touched on the device's NUMA node / on another node / wherever the main thread happens to
run, pageable and registered, pipeline threads bound and unbound; plus the price of tic_host_register itself.
Prints the node of a sample of the frames' pages (move_pages(2) with a NULL node list = query)."""
import ctypes as C, glob, os, statistics, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N

libc = C.CDLL(None, use_errno=True)
SYS_move_pages, SYS_getcpu = 279, 309

def page_nodes(arr, samples=16):
    base = arr.ctypes.data & ~4095
    n = min(samples, max(1, arr.nbytes // 4096))
    step = max(1, (arr.nbytes // 4096) // n)
    pages = (C.c_void_p * n)(*[base + 4096 * step * k for k in range(n)])
    status = (C.c_int * n)()
    rc = libc.syscall(SYS_move_pages, 0, C.c_ulong(n), pages, None, status, 0)
    return sorted(set(status)) if rc == 0 else ["move_pages failed errno %d" % C.get_errno()]

def cur_cpu_node():
    cpu, node = C.c_uint(), C.c_uint()
    libc.syscall(SYS_getcpu, C.byref(cpu), C.byref(node), None)
    return cpu.value, node.value

def parse_list(s):
    out = []
    for part in s.strip().split(","):
        if not part: continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out

nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = parse_list(open(d + "/cpulist").read())
mine = sorted(os.sched_getaffinity(0))
print("nodes:", {k: "%d cpus (%d in this process)" % (len(v), len(set(v) & set(mine))) for k, v in nodes.items()})
print("main thread on cpu %d node %d; process affinity %d cpus" % (*cur_cpu_node(), len(mine)))
L = N.load(); ctx = T.Context(0)
dn, dc = C.c_int(), C.c_int()
ctx.check(L.tic_numa_info(ctx.handle, C.byref(dn), C.byref(dc)))
print("device NUMA node %d, %d CPUs of it in this process" % (dn.value, dc.value))
h, w, n, q = 1080, 1920, 256, 50
cap = L.tic_compress_bound(h, w)
pool = np.empty((n, cap), dtype=np.uint8); pool[:] = 0
outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()

def make_frames(cpus):
    """frames allocated and first-touched by this thread while it is restricted to `cpus`"""
    if cpus: os.sched_setaffinity(0, cpus)
    fr = [np.random.default_rng(1234 + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
    blk = np.stack(fr)
    os.sched_setaffinity(0, mine)
    return fr, blk

def run(inp, bind, reps=5):
    ctx.check(L.tic_set_numa_binding(ctx.handle, bind))
    ts = []
    for r in range(reps + 1):
        t = time.perf_counter()
        ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, q, outp, caps, lens, 0))
        if r: ts.append((time.perf_counter() - t) * 1e3)
    return statistics.median(ts), min(ts)

places = [("wherever the main thread runs", None)]
for k, cp in nodes.items():
    cp = sorted(set(cp) & set(mine))
    if cp: places.append(("first-touched on node %d%s" % (k, " (the device's)" if k == dn.value else ""), cp))
for name, cpus in places:
    fr, blk = make_frames(cpus)
    print("frames %s: pages on nodes %s (separate arrays) / %s (one block)" % (name, page_nodes(fr[0]) + page_nodes(fr[n // 2]), page_nodes(blk, 64)))
    inp_p = (C.c_void_p * n)(*[f.ctypes.data for f in fr])
    for bind in (1, 0):
        print("   pageable, pipeline threads %-8s median %6.2f ms  min %6.2f" % ((("bound" if bind else "unbound"),) + run(inp_p, bind)))
    t = time.perf_counter(); ctx.check(L.tic_host_register(ctx.handle, blk.ctypes.data, blk.nbytes)); t_reg = time.perf_counter() - t
    inp_r = (C.c_void_p * n)(*[blk[i].ctypes.data for i in range(n)])
    print("   registered (tic_host_register of %d MB took %.1f ms) median %6.2f ms  min %6.2f" % ((blk.nbytes >> 20, t_reg * 1e3) + run(inp_r, 1)))
    t = time.perf_counter(); ctx.check(L.tic_host_unregister(ctx.handle, blk.ctypes.data)); print("   unregister %.1f ms" % ((time.perf_counter() - t) * 1e3))
    del fr, blk

# ---- what does registering the caller's frames cost?  (separate numpy arrays, as bench.py and most callers hold them)
fr, blk = make_frames(None)
addrs = sorted(f.ctypes.data for f in fr)
gaps = sorted(set(b - a for a, b in zip(addrs, addrs[1:])))
print("256 separate arrays: address gaps between neighbours (bytes): %s ... ; span %d MB for %d MB of pixels"
      % (gaps[:4], (addrs[-1] + h * w - addrs[0]) >> 20, (n * h * w) >> 20))
t = time.perf_counter()
ok = 0
for f in fr:
    ok += L.tic_host_register(ctx.handle, f.ctypes.data, f.nbytes) == 0
t1 = time.perf_counter() - t
print("   register 256 arrays one by one: %.2f ms (%d ok), %.1f us each" % (t1 * 1e3, ok, t1 * 1e6 / n))
inp_p = (C.c_void_p * n)(*[f.ctypes.data for f in fr])
print("   batch on them: median %6.2f ms  min %6.2f" % run(inp_p, 1))
t = time.perf_counter()
for f in fr:
    L.tic_host_unregister(ctx.handle, f.ctypes.data)
print("   unregister one by one: %.2f ms" % ((time.perf_counter() - t) * 1e3))
lo = addrs[0] & ~4095
hi = (addrs[-1] + h * w + 4095) & ~4095
t = time.perf_counter(); rc = L.tic_host_register(ctx.handle, lo, hi - lo); t1 = time.perf_counter() - t
print("   register the whole span at once: rc %d, %.2f ms" % (rc, t1 * 1e3))
if rc == 0:
    print("   batch on it: median %6.2f ms  min %6.2f" % run(inp_p, 1))
    L.tic_host_unregister(ctx.handle, lo)
for cnt in (16,):
    sub = sorted(f.ctypes.data for f in fr[:cnt])
    lo = sub[0] & ~4095; hi = (sub[-1] + h * w + 4095) & ~4095
    t = time.perf_counter(); rc = L.tic_host_register(ctx.handle, lo, hi - lo); t1 = time.perf_counter() - t
    print("   register the span of the first %d frames (%d MB): rc %d, %.2f ms" % (cnt, (hi - lo) >> 20, rc, t1 * 1e3))
    if rc == 0: L.tic_host_unregister(ctx.handle, lo)
tr = (C.c_double * 8)()
ctx.check(L.tic_compress_batch(ctx.handle, inp_p, n, h, w, w, q, outp, caps, lens, 0))
ctx.check(L.tic_last_batch_phases(ctx.handle, tr))
print("phases of a pageable batch (ms): stage %.2f enqueue %.2f slot wait %.2f | chunk wait %.2f read-back %.2f | hand-out %.2f" % (tr[0], tr[1], tr[5], tr[2], tr[3], tr[4]))
