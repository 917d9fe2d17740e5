#!/bin/bash
# usage (GPU box): tools/ab_wino_pc.sh "<ENV=val>"   - the wide Winograd shapes (tools/bench_wino.py, first section) and bench.py with the product
# kernels and with the given environment setting (e.g. TMG_WINO_PC=2: the producer-wave form for every shape), alternating
cd $GRAFT_REPO_ROOT
set=${1:-TMG_WINO_PC=2}
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "test_winograd_conv_matches_fp64 and f32" 2>&1 | tail -1
env $set python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "test_winograd_conv_matches_fp64 and f32" 2>&1 | tail -1
for i in 1 2; do
TMG_BENCH_WINO_WIDE_ONLY=1 python tools/bench_wino.py 2>/dev/null | sed 's/bf16x3.*direct/direct/' | cut -c1-70 | sed 's/^/base /'
env $set TMG_BENCH_WINO_WIDE_ONLY=1 python tools/bench_wino.py 2>/dev/null | sed 's/bf16x3.*direct/direct/' | cut -c1-70 | sed "s/^/$set /"
done
tools/ab_env.sh "$set" 3
