#!/bin/bash
# usage (GPU box): tools/ab_lib.sh A.so B.so [n] [bench args]  - bench.py with either build of the library, alternating n (3) times on one box.
# The product library is backed up first and restored on exit.
a=$1; b=$2; n=${3:-3}; shift $(( $# < 3 ? $# : 3 ))
lib=deep-turbulence_amd/libtmglow_hip.so
cp $lib $lib.ab_backup; trap 'mv -f $lib.ab_backup $lib' EXIT
mkdir -p gpurun_out; out=gpurun_out/ab_lib.txt; : > $out
for i in $(seq 1 $n); do
  for l in $a $b; do
    cp $l $lib
    python bench.py --no-cpu-baseline --no-events "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'])" >> $out
  done
done
cat $out
