#!/usr/bin/env python3
"""Register / spill table from `hipcc -Rpass-analysis=kernel-resource-usage` output: tools/regs.py <stderr.txt> [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, out = None, []
for line in txt.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}
        out.append(cur)
    for k in ['VGPRs:', 'VGPRs Spill:', 'ScratchSize [bytes/lane]:', 'Occupancy [waves/SIMD]:']:
        if k in line and cur is not None:
            cur[k] = line.split(k)[1].split()[0]
for o in out:
    n = subprocess.run(['c++filt', o['name']], capture_output=True, text=True).stdout.strip()
    if re.search(flt, n):
        print(n[:72].ljust(72), 'vgpr', o.get('VGPRs:'), 'spill', o.get('VGPRs Spill:'), 'scratch', o.get('ScratchSize [bytes/lane]:'), 'occ', o.get('Occupancy [waves/SIMD]:'))
