#!/bin/bash
# PMC counters of the ablation variants (separate passes, no tracing domains mixed in)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcA -- python3 tools/ablate.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmcB -- python3 tools/ablate.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcC -- python3 tools/ablate.py > /dev/null 2>&1
echo done
