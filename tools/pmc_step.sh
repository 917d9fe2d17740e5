#!/bin/bash
# usage (GPU box): tools/pmc_step.sh <kernel-name substring> ...   -> SQ counters (two passes) of the matching kernels in two bench steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf "$R/gpurun_out/pmc_step1" "$R/gpurun_out/pmc_step2"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$R/gpurun_out/pmc_step1" -o p -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-events > "$R/gpurun_out/pmc_step1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d "$R/gpurun_out/pmc_step2" -o p -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-events > "$R/gpurun_out/pmc_step2.log" 2>&1
cd "$R"
tail -2 gpurun_out/pmc_step1.log | cut -c1-300
tail -2 gpurun_out/pmc_step2.log | cut -c1-300
python3 tools/pmc_summary.py gpurun_out/pmc_step1 "$@"
python3 tools/pmc_summary.py gpurun_out/pmc_step2 "$@"
