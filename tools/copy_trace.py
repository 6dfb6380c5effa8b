"""Copies of a `rocprofv3 --memory-copy-trace --kernel-trace` run, largest first: direction, bytes are not in the trace, so duration and
count per direction and the longest ones.  Usage: python tools/copy_trace.py <dir>"""
import csv, glob, sys, collections
mt = glob.glob(sys.argv[1] + '/*/*_memory_copy_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(mt)):
    acc[r['Direction'].replace('MEMORY_COPY_', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in acc.items():
    v.sort(reverse=True)
    print("%-18s %5d copies, %9.1f us in all; longest: %s" % (k, len(v), sum(v), " ".join("%.0f" % x for x in v[:12])))
