#!/usr/bin/env python3
"""Per-kernel census of an assembly listing (hipcc -S --cuda-device-only, the flags of tmg_hip.build): MFMAs, global / buffer loads
and stores, `s_waitcnt vmcnt(0)` (a full drain of the vector-memory queue: in front of a store it serialises the stores behind a
load in a conditional block), 64-bit vector address additions (v_lshl_add_u64: the fp32 MFMA shares the vector ALUs), scratch.
usage: tools/isa_audit.py file.s [name-filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows, name, body = [], None, []
for ln in txt:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name, body = m.group(1), []
        continue
    if name is None:
        continue
    body.append(ln)
    if "s_endpgm" in ln:
        b = "\n".join(body)
        # vmcnt(0) directly followed (within 8 lines) by a store: the serialised-store pattern
        ser = 0
        for i, l in enumerate(body):
            if "s_waitcnt vmcnt(0)" in l and any(("_store_" in x) for x in body[i + 1:i + 9]):
                ser += 1
        rows.append((name, b.count("v_mfma"), len(re.findall(r"\b(global|buffer|flat)_load", b)), len(re.findall(r"\b(global|buffer|flat)_store", b)),
                     b.count("vmcnt(0)"), ser, b.count("v_lshl_add_u64"), b.count("scratch_")))
        name = None
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%6s %6s %6s %8s %8s %7s %7s  kernel" % ("mfma", "loads", "stores", "vmcnt(0)", "vm0>store", "add_u64", "scratch"))
for r, d in zip(rows, dem):
    if flt in d:
        print("%6d %6d %6d %8d %8d %7d %7d  %s" % (r[1:] + (d[:110],)))
