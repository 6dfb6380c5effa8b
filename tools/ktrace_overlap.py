"""How much of a rocprofv3 kernel trace has two or more kernels running at once: the span of the last `n` launches of the named kernels,
the time with >= 1 and with >= 2 of them in flight, the sum of their durations.  Usage: python tools/ktrace_overlap.py <dir> <n> <substr> [<substr> ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
n = int(sys.argv[2]); subs = sys.argv[3:]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)) if any(s in r['Kernel_Name'] for s in subs)]
rows.sort()
rows = rows[-n:]
ev = sorted([(a, 1) for a, b, _ in rows] + [(b, -1) for a, b, _ in rows])
depth = 0; last = ev[0][0]; t1 = t2 = 0
for t, d in ev:
    if depth >= 1: t1 += t - last
    if depth >= 2: t2 += t - last
    depth += d; last = t
span = rows[-1][1] - rows[0][0]
tot = sum(b - a for a, b, _ in rows)
print("%d launches over %.1f us: busy %.1f us, two or more kernels in flight %.1f us (%.0f %% of the busy time), sum of durations %.1f us; %.1f us of span per pair of launches"
      % (len(rows), span / 1e3, t1 / 1e3, t2 / 1e3, 100.0 * t2 / max(t1, 1), tot / 1e3, span / 1e3 / (len(rows) / 2)))
