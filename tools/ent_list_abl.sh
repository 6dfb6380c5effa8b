#!/bin/bash
# (Needs the symbol-list packing kernel of commit b4b7c1c, which is not in the product: check that commit out to re-run.)
# Timing builds of the symbol-list packing kernel (experiment library): 0 full, 1 no slot store, 2 no symbol rounds, 6 no list and no
# rounds, 7 nothing but loads and counts.  Usage (on the GPU box): tools/ent_list_abl.sh [content] [abls...]
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp TIC_USE_ABLATE=1 TIC_ENT_LIST=1 TIC_CONTENT=${1:-noise}
shift
for a in ${@:-0 1 2 6 7}; do
  d=gpurun_out/ent_list_abl_${TIC_CONTENT}_$a
  rm -rf "$d"
  export TIC_ENT_ABL=$a
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python tools/prof_compress_dev.py 4096 100 > "$d.txt" 2>&1
  rc=$?
  echo "ABL $a ($TIC_CONTENT): $(find "$d" -name '*kernel_stats.csv' | head -1 | xargs grep pack_list | awk -F, '{print $4}') ns"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 99; fi
done
exit 0
