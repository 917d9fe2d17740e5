#!/bin/bash
# A/B of this tree's bench.py against an older revision, alternating on ONE box.
#   here (build container):  tools/ab_old.sh prepare <rev>   - git archive <rev> into build_exp/ab_old/ and build its library there
#   on the GPU box:          tools/ab_old.sh run [n] [bench args]  -> gpurun_out/ab_old.txt
# build_exp/ is git-ignored (history stays source-only) but travels with the gpurun snapshot; pytest.ini keeps pytest out of it.
set -e
cmd=${1:-run}
if [ "$cmd" = prepare ]; then
  rev=${2:-r5-final}
  rm -rf build_exp/ab_old && mkdir -p build_exp/ab_old
  git archive "$rev" | tar -x -C build_exp/ab_old
  (cd build_exp/ab_old && python -c "import __graft_entry__ as g; g.build()")
  echo "$rev" > build_exp/ab_old/REV
  exit 0
fi
shift || true
n=${1:-3}; shift || true
mkdir -p gpurun_out; out=$PWD/gpurun_out/ab_old.txt; : > $out
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in $(seq 1 $n); do
  (cd build_exp/ab_old && python bench.py --no-cpu-baseline --no-events "$@" 2>/dev/null | python -c "$line" "old($(cat REV))" >> $out)
  python bench.py --no-cpu-baseline --no-events "$@" 2>/dev/null | python -c "$line" new >> $out
done
cat $out
