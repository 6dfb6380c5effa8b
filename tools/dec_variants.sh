#!/bin/bash
# per-kernel times of decoder builds (timing variants included: their pixels may be wrong): tools/dec_variants.sh lib1.so lib2.so ...
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so)
  rm -rf "gpurun_out/prof_$name"
  export TIC_LIB="$PWD/$lib"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/prof_$name" -- python tools/prof_decompress.py 4096 30 ${TIC_Q:-50} > "gpurun_out/prof_$name.txt" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 99; fi
  f=$(ls gpurun_out/prof_$name/*/*kernel_stats.csv | head -1)
  echo "== $name"; grep tic_decompress_dev "gpurun_out/prof_$name.txt"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dec_' in r['Name'] or 'scan_' in r['Name']:
        print("   %-40s %4s calls  %7.1f us" % (r['Name'].split('(')[0].split('::')[-1][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
