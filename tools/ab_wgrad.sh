#!/bin/bash
# usage (GPU box): tools/ab_wgrad.sh A.so B.so  - the Winograd weight-gradient micro-benchmark (tools/bench_wino.py, last section) and bench.py with
# either build of the library, alternating
cd $GRAFT_REPO_ROOT
lib=deep-turbulence_amd/libtmglow_hip.so
cp $lib $lib.ab_backup; trap 'mv -f $lib.ab_backup $lib' EXIT
for l in $1 $2; do cp $l $lib; echo "== $l"; python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "wgrad or weight_grad" 2>&1 | tail -1; python tools/bench_wino.py 2>/dev/null | sed -n '/weight gradients/,$p' | cut -c1-70; done
for i in 1 2 3; do for l in $1 $2; do cp $l $lib; python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'])"; done; done
