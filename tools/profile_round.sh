#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>     e.g. tools/profile_round.sh r2
# 1. bench JSON of the default invocation, 2. rocprofv3 kernel trace + stats of a short run, 3. HBM traffic: two PMC passes
# (FETCH_SIZE, WRITE_SIZE - they do not fit one pass; counters in their own runs, no other trace domains) folded per kernel.
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${tag}_stats" -o s -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-events > "$R/gpurun_out/${tag}_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/${tag}_pmc_fetch" -o f -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-events > "$R/gpurun_out/${tag}_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/${tag}_pmc_write" -o w -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-events > "$R/gpurun_out/${tag}_pmc_write.log" 2>&1
cd "$R"
python3 tools/traffic_json.py "gpurun_out/${tag}_pmc_fetch" "gpurun_out/${tag}_pmc_write" "gpurun_out/${tag}_traffic_per_launch.json"
cp "$(ls gpurun_out/${tag}_stats/*kernel_stats.csv | head -1)" "gpurun_out/${tag}_kernel_stats_config_M.csv"
# the bench line reads the per-launch traffic of its dominant kernel from profiles/: put this round's table there first
cp "gpurun_out/${tag}_traffic_per_launch.json" "profiles/${tag}_traffic_per_launch.json"
python3 "$R/bench.py" > "$R/gpurun_out/${tag}_bench_config_M.json" 2> "$R/gpurun_out/${tag}_bench_config_M.err"
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
t = json.load(open("gpurun_out/%s_traffic_per_launch.json" % tag))
tot = sum((v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"] for k, v in t.items() if isinstance(v, dict))
print("HBM traffic of the 2 profiled steps: %.1f GB -> %.1f GB per step (algorithmic 24.25 GB): %.2fx" % (tot / 1e9, tot / 2e9, tot / 2e9 / 24.25))
json.dump({"total_bytes_two_steps": tot, "bytes_per_step": tot / 2, "algorithmic_bytes_per_step": 24.25e9, "ratio": tot / 2 / 24.25e9},
          open("gpurun_out/%s_traffic_total.json" % tag, "w"))
PY
