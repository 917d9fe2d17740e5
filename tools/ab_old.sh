#!/bin/bash
# usage: tools/ab_old.sh [n]  - bench.py of this tree against the copy of an older tree in _old/ (git archive + its own built library), alternating
# n (3) times on one box -> gpurun_out/ab_old.txt
mkdir -p gpurun_out
out=$PWD/gpurun_out/ab_old.txt
: > $out
n=${1:-3}
for i in $(seq 1 $n); do
  (cd _old && python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old ', d['value'], d['ms_per_step'])" >> $out)
  python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'])" >> $out
done
cat $out
