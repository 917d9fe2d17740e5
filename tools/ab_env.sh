#!/bin/bash
# usage: tools/ab_env.sh "VAR=val [VAR2=val2 ...]" [n] [more settings ...]  - bench.py with the default environment and with each given
# setting (a quoted list of assignments), alternating n (3) times on one box -> gpurun_out/ab_env.txt
mkdir -p gpurun_out
out=gpurun_out/ab_env.txt
: > $out
first=$1; n=${2:-3}; shift; shift
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-40s" % sys.argv[1], d["value"], d["ms_per_step"])'
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "$line" default >> $out
  for set in "$first" "$@"; do
    [[ "$set" == *=* ]] || set="$set=1"
    env $set python bench.py --no-cpu-baseline --no-events 2>/dev/null | python -c "$line" "$set" >> $out
  done
done
cat $out
