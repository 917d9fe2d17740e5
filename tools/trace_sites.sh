#!/bin/bash
# usage (GPU box, repo root): tools/trace_sites.sh <tag> [bench args]   -> gpurun_out/<tag>_sites.txt (tools/site_times.py over a kernel trace)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf "$R/gpurun_out/${tag}_trace"
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/${tag}_trace" -o t -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-events "$@" > "$R/gpurun_out/${tag}_trace.log" 2>&1
cd "$R"
python3 tools/site_times.py "gpurun_out/${tag}_trace" --steps 3 --skip 2 --out "gpurun_out/${tag}_sites.txt" > /dev/null
# the raw trace is large: keep only the summary
rm -rf "gpurun_out/${tag}_trace"
cat "gpurun_out/${tag}_sites.txt"
