"""Durations of one kernel's launches out of a rocprofv3 kernel trace, grouped by grid size (timing experiments that launch the same
kernel with several grids).  Usage: python tools/ktrace_durations.py <dir> <kernel substring>"""
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r['Kernel_Name']:
        acc[(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for g, v in sorted(acc.items(), key=lambda kv: int(kv[0])):
    print("grid %8s: %4d launches, median %.1f us" % (g, len(v), statistics.median(v)))
