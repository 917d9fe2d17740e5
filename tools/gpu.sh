#!/bin/bash
# usage: tools/gpu.sh <gpurun-timeout-seconds> '<command to run on the GPU box>'
# Rebuilds the HIP library first and refuses to spend GPU time if the build fails.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/tmg_build.log 2>&1 || { grep -E "error" /tmp/tmg_build.log | head; echo "BUILD FAILED"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
