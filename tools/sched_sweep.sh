#!/bin/bash
# Runs tools/sched_sweep.py under each strip schedule (one process per setting: the library reads the knobs once).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
out=gpurun_out/sched_sweep.txt; : > $out
run() { echo "--- $*" >> $out; env "$@" timeout -k 10 200 python tools/sched_sweep.py 16384 >> $out 2>&1; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping" >> $out; exit 99; fi; }
run TIC_SCHED=0
run TIC_SCHED=1 TIC_CHUNK=16
run TIC_SCHED=1 TIC_CHUNK=8
run TIC_SCHED=2 TIC_CHUNK=16
run TIC_SCHED=2 TIC_CHUNK=8
run TIC_SCHED=2 TIC_CHUNK=12
cat $out
