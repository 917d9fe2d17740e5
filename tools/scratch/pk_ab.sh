#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/bench_wino.py 2>/dev/null | sed 's/bf16x3.*direct/direct/; s/(few-output kernel, fp32)//' | cut -c1-75
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"ms_per_step\"])"; done
