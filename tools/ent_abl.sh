cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 4 8 15; do
  rm -rf gpurun_out/ent_abl$a
  TIC_USE_ABLATE=1 TIC_ENT_ABL=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ent_abl$a -- python tools/prof_compress_dev.py 4096 100 > gpurun_out/ent_abl$a.txt 2>&1
  echo "ABL $a: $(find gpurun_out/ent_abl$a -name '*kernel_stats.csv' | head -1 | xargs grep pack_lane | awk -F, '{print $(NF-4)}')"
done
