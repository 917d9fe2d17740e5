#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "winograd" 2>&1 | tail -1
for i in 1 2; do TMG_BENCH_WINO_WIDE_ONLY=1 python tools/bench_wino.py 2>/dev/null | sed 's/bf16x3.*direct/direct/' | cut -c1-70; done
