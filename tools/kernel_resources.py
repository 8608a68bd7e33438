#!/usr/bin/env python3
"""Print VGPR / SGPR / LDS / scratch / occupancy per kernel of libvrt_hip (hipcc -Rpass-analysis).
    python tools/kernel_resources.py [-DFLAG ...] [name-substring]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxel_rt2_amd import build as B
rows = []
EXTRA = [a for a in sys.argv[1:] if a.startswith("-")]
ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]
for src in B.SOURCES[:2]:
    cmd = [B._hipcc()] + [f for f in B.FLAGS if f not in ("-shared",)] + EXTRA + ["-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(B.CSRC, src), "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name)}
            rows.append(cur)
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>6s} {'occ':>4s} {'vspill':>6s}")
for r in rows:
    if ONLY and not any(o in r["name"] for o in ONLY):
        continue
    print(f"{r['name'][:58]:58s} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} {r.get('TotalSGPRs',0):5d} {r.get('ScratchSize',0):8d} "
          f"{r.get('LDS Size',0):6d} {r.get('Occupancy',0):4d} {r.get('VGPRs Spill',0):6d}")
