#!/usr/bin/env python3
"""Static instruction mix per kernel of a device-only assembly listing:
   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only x.hip -o x.s ; tools/isa_mix.py x.s [name filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):\s*; @\w+\n(.*?)^\.Lfunc_end', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    cnt = collections.Counter()
    for line in body.splitlines():
        mm = re.match(r'\s+([a-z_0-9]+)\s', line)
        if not mm:
            continue
        op = mm.group(1)
        if op.startswith('v_mfma'): k = 'mfma'
        elif op.startswith('ds_'): k = op
        elif op.startswith(('global_load', 'buffer_load')): k = 'gload'
        elif op.startswith(('global_store', 'global_atomic')): k = 'gstore'
        elif op.startswith('scratch'): k = 'scratch'
        elif op.startswith(('s_load', 's_buffer')): k = 'sload'
        elif op.startswith('s_waitcnt'): k = 'waitcnt'
        elif op.startswith('s_barrier'): k = 'barrier'
        elif op.startswith('v_'): k = 'valu'
        elif op.startswith('s_'): k = 'salu'
        else: k = 'other'
        cnt[k] += 1
    print(name, dict(sorted(cnt.items(), key=lambda kv: -kv[1])))
