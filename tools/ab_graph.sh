#!/bin/bash
# eager step vs hipGraph-replayed step (bench.py --graph), alternating on one box; -> gpurun_out/ab_graph.txt
mkdir -p gpurun_out
out=gpurun_out/ab_graph.txt
: > $out
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager', d['value'], d['ms_per_step'])" >> $out
  python bench.py --no-cpu-baseline --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph', d['value'], d['ms_per_step'])" >> $out
done
cat $out
