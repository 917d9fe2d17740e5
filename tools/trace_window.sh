#!/bin/bash
# usage (GPU box, repo root): tools/trace_window.sh <tag>  -> kernel-trace site table of the trainer's 10-step BPTT windows (per time-step)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf "$R/gpurun_out/${tag}_wtrace"
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/${tag}_wtrace" -o t -- python3 "$R/tools/window_run.py" --windows 2 --adam hip "$@" > "$R/gpurun_out/${tag}_wtrace.log" 2>&1
cd "$R"
tail -1 "gpurun_out/${tag}_wtrace.log" | cut -c1-400
# two calls of trainParallel x 2 windows = 4 optimizer steps: skip the warm-up call's two
python3 tools/site_times.py "gpurun_out/${tag}_wtrace" --steps 2 --skip 2 --per 10 --out "gpurun_out/${tag}_window_sites.txt" | head -70
rm -rf "gpurun_out/${tag}_wtrace"
