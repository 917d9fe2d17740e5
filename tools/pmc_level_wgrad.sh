#!/bin/bash
# usage (GPU box): tools/pmc_level_wgrad.sh  -> SQ counters of level_wgrad_kernel and the three kernels it replaces (tools/bench_level_wgrad.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/pmc_lwa" -o p -- python3 "$R/tools/bench_level_wgrad.py" > "$R/gpurun_out/pmc_lwa.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d "$R/gpurun_out/pmc_lwb" -o p -- python3 "$R/tools/bench_level_wgrad.py" > "$R/gpurun_out/pmc_lwb.log" 2>&1
cd "$R"
python3 - <<'PY'
import csv, collections, glob
for d in ("pmc_lwa", "pmc_lwb"):
    fs = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:40]
        if not any(s in k for s in ("level_wgrad", "wgrad_thin", "mix_wgrad", "conv_wgrad_kernel")):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        wc = v.get("SQ_WAVE_CYCLES", 1)
        print(d, k, "wave cycles %.3g" % wc)
        for name, val in sorted(v.items()):
            if name != "SQ_WAVE_CYCLES":
                print("   %-26s %8.3f per wave cycle" % (name, val / wc))
PY
