#!/bin/bash
# usage (GPU box, repo root): tools/prof_bf3.sh -> gpurun_out/r5_kernel_stats_config_M_bf16x3.csv (the step with the bf16x3 switch, rocprofv3 kernel stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp TMG_WINO_BF3=1
cd /tmp
rm -rf "$R/gpurun_out/bf3_stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/bf3_stats" -o s -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-events > "$R/gpurun_out/bf3_stats.log" 2>&1
cd "$R"
cp "$(ls gpurun_out/bf3_stats/*kernel_stats.csv | head -1)" gpurun_out/r5_kernel_stats_config_M_bf16x3.csv
tail -1 gpurun_out/bf3_stats.log | cut -c1-400
