#!/bin/bash
# round-6 artefacts on the GPU box: bench JSON, kernel stats, PMC traffic, launch sites, smoke
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?"
tools/profile_round.sh r6 2>&1 | tail -5
tools/trace_sites.sh r6 > /dev/null 2>&1
head -3 gpurun_out/r6_sites.txt
cat gpurun_out/r6_bench_config_M.json | cut -c1-600
