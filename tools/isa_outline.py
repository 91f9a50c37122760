#!/usr/bin/env python3
"""Outline of one kernel's ISA: memory instructions, waits, barriers, MFMA runs (counts), branches.
usage: hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only -I ga3c_amd/csrc -o /tmp/eng.s ga3c_amd/csrc/ga3c_engine.hip
       tools/isa_outline.py /tmp/eng.s <mangled-name substring> [max lines]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and sys.argv[2] in l.split(':')[0])
out, cnt = [], 0
for l in lines[start + 1:]:
    if 's_endpgm' in l:
        break
    if 'v_mfma' in l:
        cnt += 1
        continue
    if re.search(r'global_load|global_store|s_waitcnt vmcnt|s_barrier|scratch_', l):
        if cnt:
            out.append('   [%d mfma]' % cnt)
            cnt = 0
        out.append(l)
if cnt:
    out.append('   [%d mfma]' % cnt)
print('\n'.join(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 200]))
