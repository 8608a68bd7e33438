#!/usr/bin/env python3
"""List the loops of one kernel in a hipcc -S listing with their static instruction counts.

usage: isa_loops.py k.s '_ZN3vrt8k_renderILb0ELb0EE' [min_instr]
(k.s from: hipcc <build flags> --cuda-device-only -S voxel_rt2_amd/csrc/vrt_kernels.hip -o k.s)
"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
prefix = sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = [i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm')][0]
body = lines[start:end + 1]
def isinstr(l):
    s = l.strip()
    return bool(s) and not s.startswith(';') and not s.startswith('.') and not s.split(';')[0].strip().endswith(':')
print('kernel instructions', sum(isinstr(l) for l in body))
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r'(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        j = labels[m.group(2)]
        cnt = sum(isinstr(x) for x in body[j:i + 1])
        if cnt >= floor: print(f'{m.group(2):12s} lines {start+j:6d}-{start+i:6d}  {cnt:5d} instr')
