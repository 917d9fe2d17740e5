#!/bin/bash
# usage (GPU box, repo root): tools/window_stats.sh <tag>  -> rocprofv3 --kernel-trace --stats of the trainer's BPTT windows (tools/window_run.py,
# two calls of two windows each: divide by 40 time-steps) -> gpurun_out/<tag>_kernel_stats_window.csv
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf "$R/gpurun_out/${tag}_wstats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${tag}_wstats" -o w -- python3 "$R/tools/window_run.py" --windows 2 "$@" > "$R/gpurun_out/${tag}_wstats.log" 2>&1
cd "$R"
cp "$(ls gpurun_out/${tag}_wstats/*kernel_stats.csv | head -1)" "gpurun_out/${tag}_kernel_stats_window.csv"
rm -rf "gpurun_out/${tag}_wstats"
tail -1 "gpurun_out/${tag}_wstats.log" | cut -c1-200
head -12 "gpurun_out/${tag}_kernel_stats_window.csv" | cut -c1-160
