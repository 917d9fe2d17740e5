#!/bin/bash
# usage (GPU box): tools/ab_lib.sh A.so B.so [n] [bench args]  - bench.py with either build of the library, alternating n (3) times on one box
a=$1; b=$2; n=${3:-3}; shift 3
mkdir -p gpurun_out; out=gpurun_out/ab_lib.txt; : > $out
for i in $(seq 1 $n); do
  for l in $a $b; do
    cp $l deep-turbulence_amd/libtmglow_hip.so
    python bench.py --no-cpu-baseline --no-events "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'])" >> $out
  done
done
cat $out
