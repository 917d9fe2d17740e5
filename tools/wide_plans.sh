#!/bin/bash
# tools/bench_wide.py under a list of conv_fwd launch plans (TMG_FWD_PLAN=MT,WM,WN,NTW,GMUL,KCHMAX); output -> gpurun_out/wide_plans.txt
mkdir -p gpurun_out
out=gpurun_out/wide_plans.txt
: > $out
python tools/bench_wide.py >> $out 2>&1
for plan in "$@"; do
  TMG_FWD_PLAN=$plan timeout 300 python tools/bench_wide.py >> $out 2>&1
done
grep -c . $out
