#!/usr/bin/env python
"""kernel resource usage of one .hip file (VGPRs, SGPRs, scratch, LDS, occupancy per kernel) from hipcc's
-Rpass-analysis=kernel-resource-usage.  usage: tools/kres.py pixell_amd/csrc/leg_s0.hip [extra hipcc flags]"""
import subprocess, sys, re
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]+sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
	m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
	if not m: continue
	k, v = m.group(1), m.group(2)
	if k == "Function Name":
		cur = {"name": v}; rows.append(cur)
	elif cur is not None: cur[k.split(" [")[0]] = v
for r in rows:
	name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
	name = re.sub(r"\(.*", "", name)
	print("%-60s VGPR %4s SGPR %4s scratch %3s LDS %6s occ %s" % (name[:60], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
