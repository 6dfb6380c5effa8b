#!/bin/bash
# (Needs the symbol-list packing kernel of commit b4b7c1c, which is not in the product: check that commit out to re-run.)
# Packing kernels of the device entropy stage side by side: 8 lanes per block (default) against the symbol list, noise and Lenna,
# per-kernel times by rocprofv3.  Usage (on the GPU box): tools/ent_list_ab.sh [dim] [reps]
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
dim=${1:-4096}; reps=${2:-200}
for content in noise lenna; do
  for kind in 0 1; do
    d=gpurun_out/ent_list_${content}_$kind
    rm -rf "$d"
    export TIC_ENT_LIST=$kind TIC_CONTENT=$content
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python tools/prof_compress_dev.py "$dim" "$reps" > "$d.txt" 2>&1
    rc=$?
    echo "== $content kernel $kind (exit $rc)"; grep tic_compress_dev "$d.txt"
    python - "$d" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "entropy" in r["Name"] or "dctq" in r["Name"]:
            print("   %-60s calls %5s avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 99; fi
  done
done
exit 0
