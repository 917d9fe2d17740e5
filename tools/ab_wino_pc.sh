#!/bin/bash
# usage (GPU box): tools/ab_wino_pc.sh   - the wide Winograd shapes with the product kernel and with the producer-wave form (TMG_WINO_PC=1)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "test_winograd_conv_matches_fp64 and f32" 2>&1 | tail -3
TMG_WINO_PC=1 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "test_winograd_conv_matches_fp64 and f32" 2>&1 | tail -5
for i in 1 2; do
TMG_BENCH_WINO_WIDE_ONLY=1 python tools/bench_wino.py 2>/dev/null | sed 's/bf16x3.*direct/direct/' | sed 's/^/base /'
TMG_WINO_PC=1 TMG_BENCH_WINO_WIDE_ONLY=1 python tools/bench_wino.py 2>/dev/null | sed 's/bf16x3.*direct/direct/' | sed 's/^/pc   /'
done
