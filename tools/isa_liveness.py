#!/usr/bin/env python
"""VGPR liveness of a straight-line stretch of gfx950 assembly (hipcc -S): for a line range of the .s file, the number of live
vector registers before every instruction and, at the point of maximum pressure (or at --at LINE), where each live register
was defined.  Control flow is ignored (meant for the unrolled bodies of the line-FFT kernels).
usage: tools/isa_liveness.py file.s first last [--at LINE]"""
import re, sys
def regs(tok):
	out = []
	for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
		if m.group(3) is not None: out.append(int(m.group(3)))
		else: out += list(range(int(m.group(1)), int(m.group(2)) + 1))
	return out
NODEF = ('ds_write', 'scratch_store', 'global_store', 'buffer_store', 'flat_store', 's_', 'v_cmp', 'v_readlane', 'v_readfirstlane', 'global_atomic', 'ds_add')
ACC = ('v_fmac', 'v_mac', 'v_writelane', 'v_accvgpr')
def parse(line):
	l = line.split(';')[0].strip()
	if not l or l.startswith('.') or l.endswith(':'): return None
	parts = l.split(None, 1)
	op = parts[0]; ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
	if op.startswith(NODEF) and not op.startswith('s_waitcnt'): return op, [], sum((regs(o) for o in ops), [])
	if op.startswith('s_'): return op, [], []
	d = regs(ops[0]) if ops else []
	u = sum((regs(o) for o in ops[1:]), [])
	if op.startswith(ACC): u += d
	return op, d, u
if __name__ == "__main__":
	f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
	at = int(sys.argv[sys.argv.index('--at') + 1]) if '--at' in sys.argv else None
	lines = open(f).read().split('\n')
	ins = [(i + 1, parse(lines[i])) for i in range(a - 1, b)]
	ins = [(n, p) for n, p in ins if p]
	live = set(); before = {}
	for n, (op, d, u) in reversed(ins):
		live -= set(d); live |= set(u); before[n] = set(live)
	mx = max(before, key=lambda n: len(before[n]))
	tgt = at if at in before else mx
	print("live-in at start: %d, max %d at line %d, at line %d: %d" % (len(before[ins[0][0]]), len(before[mx]), mx, tgt, len(before[tgt])))
	lastdef = {}
	for n, (op, d, u) in ins:
		if n >= tgt: break
		for r in d: lastdef[r] = (n, op)
	groups = {}
	for r in sorted(before[tgt]):
		groups.setdefault(lastdef.get(r, (0, 'live-in')), []).append(r)
	for (n, op), rs in sorted(groups.items()): print("  line %6d %-24s %s" % (n, op, ' '.join('v%d' % r for r in rs)))
