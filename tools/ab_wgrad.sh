#!/bin/bash
# usage (GPU box): tools/ab_wgrad.sh "<ENV=1>"  - the Winograd weight-gradient micro-benchmark (tools/bench_wino.py, last section) and bench.py with
# the default build and with the given environment setting, alternating
cd $GRAFT_REPO_ROOT
set=$1
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "wgrad or weight_grad" 2>&1 | tail -1
echo "== default"; python tools/bench_wino.py 2>/dev/null | sed -n '/weight gradients/,$p' | cut -c1-70
echo "== $set"; env $set python tools/bench_wino.py 2>/dev/null | sed -n '/weight gradients/,$p' | cut -c1-70
tools/ab_env.sh "$set" 3
