#!/bin/bash
# Runs a list of GPU steps on the gpurun box, each under its own timeout; stops at the first step that is killed
# by its timeout (never starts another GPU step after a hang).  Usage: tools/gpu_run.sh step1 step2 ...
# Steps: microbench[2..7] | tests | tests_fast | tests_all | smoke | bench | bench_cold | bench_exact | bench16k | sweep | content |
#        width_sweep | shape_ab | prof | prof_cold | pmc_rd | pmc_wr | pmc_sq | pmc_sq2 | pmc_cal | pmc_cal_wr | config5 | ...   (binaries: make -C tools)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # name timeout cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.txt" 2>&1
  local rc=$?
  echo "=== $name exit $rc"; tail -n 25 "gpurun_out/$name.txt" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit 99; fi
  return 0
}
for step in "$@"; do
  case $step in
    microbench) run microbench 240 ./tools/bin/microbench ;;
    microbench2) run microbench2 240 ./tools/bin/microbench2 ;;
    microbench3) run microbench3 240 ./tools/bin/microbench3 ;;
    microbench4) run microbench4 300 ./tools/bin/microbench4 ;;
    microbench5) run microbench5 300 ./tools/bin/microbench5 ;;
    microbench6) run microbench6 300 ./tools/bin/microbench6 ;;
    microbench7) run microbench7 300 ./tools/bin/microbench7 ;;
    microbench8) run microbench8 300 ./tools/bin/microbench8 ;;
    compress_dev) run compress_dev 200 python tools/prof_compress_dev.py 4096 200 ;;
    prof_cdev)  rm -rf gpurun_out/prof_cdev; run prof_cdev 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cdev -- python tools/prof_compress_dev.py 4096 200 ;;
    stress)     run stress 900 python tools/stress_parity.py ${TIC_STRESS_ITERS:-300} ;;
    decomp)     run decomp 200 python tools/prof_decompress.py 4096 30 50; run decomp90 200 python tools/prof_decompress.py 4096 30 90; export TIC_CONTENT=lenna; run decomp_lenna 200 python tools/prof_decompress.py 4096 30 50; unset TIC_CONTENT ;;
    prof_decomp) rm -rf gpurun_out/prof_decomp; run prof_decomp 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_decomp -- python tools/prof_decompress.py 4096 50 50 ;;
    pmc_decomp) rm -rf gpurun_out/pmc_decomp; run pmc_decomp 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_decomp -- python tools/prof_decompress.py 4096 5 50 ;;
    pmc_decomp2) rm -rf gpurun_out/pmc_decomp2; run pmc_decomp2 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_decomp2 -- python tools/prof_decompress.py 4096 5 50 ;;
    stress_ent) run stress_ent 600 python tools/stress_entropy.py ${TIC_STRESS_ENT:-300} ;;
    decomp16k)  run decomp16k 300 python tools/prof_decompress.py 16384 5 50 ;;
    stress_dec) run stress_dec 600 python tools/stress_decoder.py ${TIC_STRESS_DEC:-150} ;;
    content)    run content 300 python tools/natural_content.py ${TIC_CONTENT_LIBS:-} ;;
    width_sweep) run width_sweep 400 python tools/width_sweep.py 134 1024 4096 64 ;;
    shape_ab)   run shape_ab 400 python tools/shape_ab.py ;;
    ab_cold)    run ab_cold 400 python tools/ab_cold.py ;;
    tests)      run pytest_gpu 900 python -m pytest tests -m gpu -x -q ;;
    tests_fast) run pytest_gpu_fast 600 python -m pytest tests -m gpu -x -q -k "not 16384" ;;
    tests_all)  run pytest_gpu_all 900 python -m pytest tests -m gpu -q ;;
    smoke)      run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)      run bench 300 python bench.py ;;
    bench_cold) run bench_cold 300 python bench.py --steps 50 --warmup 10 --settle-ms 1 --no-cpu-baseline --no-config4 ;;
    bench_exact) run bench_exact 300 python bench.py --variant exact --no-cpu-baseline ;;
    bench16k)   run bench16k 300 python bench.py --height 16384 --width 16384 --steps 20 --warmup 3 --no-cpu-baseline ;;
    config5)    for q in 10 50 90; do
                  rm -rf gpurun_out/c5_prof_q$q gpurun_out/c5_rd_q$q gpurun_out/c5_wr_q$q
                  C5="bench.py --height 16384 --width 16384 --quality $q --steps 20 --warmup 3 --no-cpu-baseline --no-config4 --no-cold"
                  run c5_cold_q$q 300 python bench.py --height 16384 --width 16384 --quality $q --steps 20 --warmup 3 --no-cpu-baseline --no-config4
                  run c5_prof_q$q 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5_prof_q$q -- python $C5
                  run c5_rd_q$q 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/c5_rd_q$q -- python bench.py --height 16384 --width 16384 --quality $q --steps 3 --warmup 1 --settle-ms 1 --no-cpu-baseline --no-config4 --no-cold
                  run c5_wr_q$q 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/c5_wr_q$q -- python bench.py --height 16384 --width 16384 --quality $q --steps 3 --warmup 1 --settle-ms 1 --no-cpu-baseline --no-config4 --no-cold
                done ;;
    driver_flags) T=${TIC_TAG:-a}
                run df_tool_$T 300 python tools/driver_flags.py 5
                for i in 1 2 3; do run df_short_${T}$i 200 python3 bench.py --gpus 1 --steps 20 --warmup 5; done
                run df_long_$T 300 python3 bench.py --gpus 1 --steps 5000 --warmup 2000 --no-cpu-baseline ;;
    clock_profile) run clock_profile 120 python tools/clock_profile.py 500 300 ;;
    bench_set)  run bench_set 300 python tools/bench_set_timing.py ;;
    tests_dec)  run pytest_gpu_dec 900 python -m pytest tests -m gpu -x -q -k "decompress or decoder or benchmark_set or truncated or scaled or config5_frame or shipped" ;;
    stress_rounds) export TIC_DECODE_ROUNDS=32; run stress_dec_r32 600 python tools/stress_decoder.py ${TIC_STRESS_DEC:-300}; unset TIC_DECODE_ROUNDS ;;
    small_probe) run small_probe 120 python tools/small_compress_probe.py ;;
    ab_r4)      run ab_r4 400 python tools/ab_libs.py tools/bin/lib_r4.so --rounds 7; run ab_r4_q90 300 python tools/ab_libs.py tools/bin/lib_r4.so --rounds 5 --quality 90 ;;
    kernarg)    export HIP_FORCE_DEV_KERNARG=0; run kernarg0 120 python tools/driver_flags.py 3; export HIP_FORCE_DEV_KERNARG=1; run kernarg1 120 python tools/driver_flags.py 3; unset HIP_FORCE_DEV_KERNARG ;;
    driver_prof) rm -rf gpurun_out/df_prof; run df_prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/df_prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-config4 ;;
    sweep)      run sweep 600 python tools/sweep.py ;;
    prof)       rm -rf gpurun_out/prof; run prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --no-cpu-baseline --no-cold --no-config4 ;;
    prof_cold)  rm -rf gpurun_out/prof_cold; run prof_cold 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cold -- python bench.py --steps 50 --warmup 10 --settle-ms 1 --no-cpu-baseline --no-config4 ;;
    pmc_rd)     rm -rf gpurun_out/pmc_rd; run pmc_rd 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rd -- python bench.py --steps 5 --warmup 1 --settle-ms 1 --no-cpu-baseline --no-cold --no-config4 ;;
    pmc_wr)     rm -rf gpurun_out/pmc_wr; run pmc_wr 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_wr -- python bench.py --steps 5 --warmup 1 --settle-ms 1 --no-cpu-baseline --no-cold --no-config4 ;;
    pmc_sq)     rm -rf gpurun_out/pmc_sq; run pmc_sq 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -- python bench.py --steps 5 --warmup 1 --settle-ms 1 --no-cpu-baseline --no-cold --no-config4 ;;
    pmc_ent)    rm -rf gpurun_out/pmc_ent gpurun_out/pmc_ent2; run pmc_ent 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_ent -- python tools/prof_compress_dev.py 4096 10
                run pmc_ent2 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_ent2 -- python tools/prof_compress_dev.py 4096 10 ;;
    pmc_sq2)    rm -rf gpurun_out/pmc_sq2; run pmc_sq2 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_sq2 -- python bench.py --steps 5 --warmup 1 --settle-ms 1 --no-cpu-baseline --no-cold --no-config4 ;;
    pmc_lds)    rm -rf gpurun_out/pmc_lds; run pmc_lds 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_lds -- ./tools/bin/microbench4 ;;
    pmc_cal)    rm -rf gpurun_out/pmc_cal; run pmc_cal 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cal -- ./tools/bin/microbench ;;
    pmc_cal_wr) rm -rf gpurun_out/pmc_cal_wr; run pmc_cal_wr 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_cal_wr -- ./tools/bin/microbench ;;
    *) echo "unknown step $step" ;;
  esac
done
exit 0
