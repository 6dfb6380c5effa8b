#!/bin/bash
# Stall study of the hybrid kernel's timing variants: one rocprofv3 --pmc pass per counter group.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
DIM=${1:-4096}
mkdir -p gpurun_out/pmcs
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU" ; do
  i=$((i+1))
  rm -rf gpurun_out/pmcs/g$i
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcs/g$i -- python tools/run_variants.py $DIM 3 > gpurun_out/pmcs/g$i.log 2>&1
  rc=$?; echo "group $i rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 99; fi
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcs/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "dctq_hybrid" not in k: continue
        name = k[k.index("dctq_hybrid"):][:32]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_study.txt", "w") as out:
    for name in sorted(acc):
        out.write(name + "\n")
        for c in sorted(acc[name]):
            v = acc[name][c]
            out.write("   %-34s %16.0f  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open("gpurun_out/pmc_study.txt").read())
PY
