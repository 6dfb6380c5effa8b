"""Timeline of the LAST tic_compress_batch call of a `rocprofv3 --kernel-trace --memory-copy-trace -- python tools/prof_batch.py` run:
when the uploads run and where they pause, when the streams come back, which kernels run.  Usage: python tools/batch_timeline.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
kt = glob.glob(d + '/*/*_kernel_trace.csv')[0]
mt = glob.glob(d + '/*/*_memory_copy_trace.csv')[0]
ev = []
for r in csv.DictReader(open(kt)):
    name = r['Kernel_Name']
    for key in ("readback_rows", "entropy_pack", "entropy_place", "entropy_tilesum", "dctq_strip", "copyBuffer", "fillBuffer"):
        if key in name: name = key
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', name[:32]))
for r in csv.DictReader(open(mt)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r['Direction'].replace('MEMORY_COPY_', '')))
ev.sort()
up = [e for e in ev if e[3] == 'HOST_TO_DEVICE' and e[1] - e[0] > 20000]  # the frames (the small ones are tables and lengths)
calls = [[up[0]]]
for a, b in zip(up, up[1:]):
    if b[0] - a[1] > 900_000: calls.append([])
    calls[-1].append(b)
c = calls[-1]
t0 = c[0][0]
end = max(e[1] for e in ev if e[0] >= t0)
print("calls seen (frames uploaded in each): %s; the last one: uploads from 0 to %.2f ms, last event ends at %.2f ms" % ([len(x) for x in calls], (c[-1][1] - t0) / 1e6, (end - t0) / 1e6))
busy, ce = 0, t0
for s, e, _, _ in c:
    s = max(s, ce)
    if e > s: busy += e - s; ce = e
print("upload link busy %.2f ms (%d copies, %.1f us each on average, up to two in flight)" % (busy / 1e6, len(c), sum(e[1] - e[0] for e in c) / len(c) / 1e3))
gaps = [(b[0] - max(x[1] for x in c[:i + 1]), (a[1] - t0) / 1e6, i + 1) for i, (a, b) in enumerate(zip(c, c[1:]))]
big = [g for g in gaps if g[0] > 30000]
print("pauses of the uploads > 30 us: %d, %.2f ms in all:" % (len(big), sum(g[0] for g in big) / 1e6), ", ".join("%.0f us after frame %d (%.2f ms)" % (g[0] / 1e3, g[2], g[1]) for g in big))
for kind in sorted(set(e[3] for e in ev if e[2] == 'C' and e[0] >= t0)):
    iv = [e for e in ev if e[2] == 'C' and e[3] == kind and e[0] >= t0 and e[1] - e[0] > 20000]
    if kind != 'HOST_TO_DEVICE' and iv:
        print("copies %s: %d, at ms %s" % (kind, len(iv), " ".join("%.2f(+%.2f)" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6) for e in iv)))
for nm in sorted(set(e[3] for e in ev if e[2] == 'K' and e[0] >= t0)):
    iv = [e for e in ev if e[2] == 'K' and e[3] == nm and e[0] >= t0]
    print("kernel %-18s %3d launches, %7.1f us each, first at %.2f ms, last ends at %.2f ms" % (nm, len(iv), sum(e[1] - e[0] for e in iv) / len(iv) / 1e3, (iv[0][0] - t0) / 1e6, (max(e[1] for e in iv) - t0) / 1e6))
