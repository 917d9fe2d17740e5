#!/bin/bash
# usage (GPU box): tools/pmc_wino3.sh  -> SQ counters of the wide Winograd kernels (fp32 MFMA and the bf16x3 variant) on the micro-benchmark shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp TMG_BENCH_WINO_WIDE_ONLY=1
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/pmc_wino3a" -o p -- python3 "$R/tools/bench_wino.py" > "$R/gpurun_out/pmc_wino3a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d "$R/gpurun_out/pmc_wino3b" -o p -- python3 "$R/tools/bench_wino.py" > "$R/gpurun_out/pmc_wino3b.log" 2>&1
cd "$R"
python3 - <<'PY'
import csv, collections, glob
for d in ("pmc_wino3a", "pmc_wino3b"):
    fs = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:28]
        if "wino_fwd" not in k:
            continue
        # only the gate-conv launches (grid 256 x 2 x 512 threads = largest): keep all, normalised per wave cycle
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        wc = v.get("SQ_WAVE_CYCLES", 1)
        print(d, k)
        for name, val in sorted(v.items()):
            if name != "SQ_WAVE_CYCLES":
                print("   %-26s %8.3f per wave cycle" % (name, val / wc))
PY
