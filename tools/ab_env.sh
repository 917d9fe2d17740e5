#!/bin/bash
# usage: tools/ab_env.sh VAR [n]  - bench.py with and without VAR=1 in the environment, alternating n (3) times on one box -> gpurun_out/ab_env.txt
mkdir -p gpurun_out
out=gpurun_out/ab_env.txt
: > $out
var=$1; n=${2:-3}
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default   ', d['value'], d['ms_per_step'])" >> $out
  env $var=1 python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=1', d['value'], d['ms_per_step'])" >> $out
done
cat $out
