#!/bin/bash
# Per-kernel times of the device decoder over forced range lengths (TIC_DECODE_RANGE, a test hook): [TIC_DIM=512] tools/range_sweep.sh <quality> <content> r1 r2 ...
export TMPDIR=/tmp TIC_TEST_HOOKS=1
q=$1; c=$2; shift 2
[ "$c" = lenna ] && export TIC_CONTENT=lenna
for r in "$@"; do
  export TIC_DECODE_RANGE=$r
  rm -rf gpurun_out/sw_$r
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw_$r -- python3 tools/prof_decompress.py ${TIC_DIM:-4096} 20 $q > gpurun_out/sw_$r.txt 2>&1 || exit 1
  echo "== q=$q $c range $r: $(grep tic_decompress_dev gpurun_out/sw_$r.txt | sed 's/.*: //')"
  python tools/kstats.py gpurun_out/sw_$r | grep -v "^##" | awk '{printf "%s %s | ", $NF=="us"?$(NF-1):"", substr($2,1,28)} END {print ""}'
done
