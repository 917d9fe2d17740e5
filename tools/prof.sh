#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench.py arguments]
# rocprofv3 kernel trace + stats of a short bench.py run -> gpurun_out/<tag>/*.csv, then the per-step top kernels.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$tag" -o "$tag" -- python3 "$R/bench.py" --no-cpu-baseline --no-events "$@" > "$R/gpurun_out/$tag.log" 2>&1
cd "$R"
python3 tools/prof_top.py "gpurun_out/$tag" "$@"
