#!/usr/bin/env python3
"""Launch sites of one step from a rocprofv3 kernel trace: kernels aggregated by (name, grid size, LDS size) - the grid tells the
launch sites of one kernel apart (level, direction) - with launches and time per step, VGPR / AGPR / LDS of the code object, and
the share of the step.  Also the busy time per phase of the step (encoder / level k forward / level k backward) cut at the
checker_kernel / checker4_kernel launches that open and close a level.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-events
  python tools/site_times.py gpurun_out/trace [--steps 3 --skip 2] [--out profiles/r4_sites_config_M.txt]
"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    if name.startswith("at::native") or "at::native" in name:
        m = re.search(r"(vectorized_elementwise_kernel|elementwise_kernel\w*|CatArrayBatchedCopy\w*|reduce_kernel|index\w*|fill\w*)", name)
        f = re.search(r"(CUDAFunctor_\w+|\w+Functor\w*|direct_copy_kernel_cuda|FillFunctor)", name)
        return "torch:" + (m.group(1) if m else name[:40]) + (":" + f.group(1) if f else "")
    return name[:64]


def main():
    d = sys.argv[1]
    steps, skip, out, per, phases = 3, 2, None, 1, True
    a = sys.argv[2:]
    for i, v in enumerate(a):
        if v == "--steps":
            steps = int(a[i + 1])        # optimizer steps to aggregate
        if v == "--skip":
            skip = int(a[i + 1])         # optimizer steps to skip first
        if v == "--out":
            out = a[i + 1]
        if v == "--per":
            per = int(a[i + 1])          # model time-steps per optimizer step (a BPTT window: 10): the table is per time-step
            phases = False
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # steps are delimited by the optimizer launch
    ends = [i for i, r in enumerate(rows) if "adam_step_kernel" in r["Kernel_Name"]]
    assert len(ends) >= skip + steps, (len(ends), skip, steps)
    lo = ends[skip - 1] + 1 if skip > 0 else 0
    hi = ends[skip + steps - 1] + 1
    sel = rows[lo:hi]
    agg = {}
    tot = 0.0
    for r in sel:
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1),
               int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1), int(r["LDS_Block_Size"]))
        e = agg.setdefault(key, [0, 0.0, r])
        e[0] += 1
        e[1] += dt
        tot += dt
    wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
    nopt = steps
    steps = steps * per
    lines = ["%d optimizer step(s) of %d time-step(s): kernel time %.2f ms / time-step in %.0f launches, wall %.2f ms / time-step (GPU idle %.1f %%)" % (
        nopt, per, tot / steps / 1e3, len(sel) / steps, wall / steps / 1e3, 100.0 * (1 - tot / wall)),
        "%9s %6s %9s  %-64s %-16s %6s %5s %5s" % ("ms/step", "n/step", "us/launch", "kernel", "grid (blocks)", "LDS", "VGPR", "AGPR")]
    for key, (n, t, r) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t / steps < 20.0:
            continue
        lines.append("%9.3f %6.1f %9.1f  %-64s %-16s %6d %5s %5s" % (t / steps / 1e3, n / steps, t / n, key[0], "%dx%dx%d" % key[1:4], key[4],
                                                                    r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?")))
    if not phases:
        txt = "\n".join(lines)
        print(txt)
        if out:
            open(out, "w").write(txt + "\n")
        return
    # ---- phases of one step (the last selected one): the level boundaries are the checker launches
    s0 = ends[skip + steps - 2] + 1
    one = rows[s0:hi]
    cuts = [i for i, r in enumerate(one) if "checker_kernel" in r["Kernel_Name"] or "checker4_kernel" in r["Kernel_Name"]]
    nlev = len(cuts) // 2
    lines.append("")
    lines.append("busy time between the level boundaries of one step (forward: a level ends with its un-squeeze, backward: it starts with the")
    lines.append("adjoint of that launch; the deepest level runs first in the generative direction):")
    edges = [0] + [c + 1 for c in cuts[:nlev]] + [c for c in cuts[nlev:]] + [len(one)]
    labels = (["encoder fwd + level %d fwd" % nlev] + ["level %d fwd" % (nlev - k) for k in range(1, nlev)] + ["loss"]
              + ["level %d bwd" % (k + 1) for k in range(nlev - 1)] + ["level %d bwd + encoder bwd + Adam" % nlev])
    for (a_, b_), lb in zip(zip(edges[:-1], edges[1:]), labels):
        part = one[a_:b_]
        if not part:
            continue
        busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in part) / 1e6
        lines.append("  %-40s %7.2f ms  %5d launches" % (lb, busy, len(part)))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
