"""Instruction mix of the big basic blocks of one kernel in an ISA listing (hipcc -S --cuda-device-only).
usage: python tools/isa_blocks.py file.s <mangled-name-substring> [min_fma]"""
import sys, re, collections
lines = open(sys.argv[1]).read().split("\n"); key = sys.argv[2]; minf = int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E") )
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks = []; cur = ["entry", []]
for l in lines[start+1:end]:
	if l.startswith(".LBB"): blocks.append(cur); cur = [l.split(":")[0], []]
	elif l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"): cur[1].append(l.split()[0])
blocks.append(cur)
for name, ins in blocks:
	c = collections.Counter(ins)
	if sum(v for k, v in c.items() if k.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64"))) < minf: continue
	grp = collections.Counter()
	for k, v in c.items():
		g = ("fma64" if k.split("_e")[0] in ("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64") else "swap" if "permlane" in k else "lds" if k.startswith("ds_") else "smem" if k.startswith("s_load") else
		     "vmem" if k.startswith("global_") or k.startswith("buffer_") else "wait" if k.startswith("s_waitcnt") or k == "s_nop" else "salu" if k.startswith("s_") else "valu32")
		grp[g] += v
	print(name, len(ins), dict(grp))
	print("   ", {k: v for k, v in c.most_common(14)})
