# round 6: the theta planner's shared modulus g off the BASELINE sizes -- for every grid of tools/chain_lab.py's o1..o5 the planner's own choice against every
# modulus it could have taken (lab build: PXS_THETA_G forces it; tools/build_variants.sh lab "-DPXS_LAB" <all stems>, copied to tools/libpxsht_lab.so for the call)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_planner_sweep; mkdir -p $O; : > $O/sweep_${TAG:-a}.txt
export PIXELL_AMD_LIB=tools/libpxsht_lab.so
for cfg in ${CFGS:-o1 o2 o3 o4 o5}; do
	gs=$(python - $cfg <<'PY'
import sys, re
sys.path.insert(0, "tools"); src = open("tools/chain_lab.py").read()
cfg = sys.argv[1]
m = re.search(r'"%s": \(\((\d+), (\d+)\), (\d+)' % cfg, src); ny, nx, lmax = (int(v) for v in m.groups())
N = 2*ny
def smooth(n, ps):
	for p in ps:
		while n % p == 0: n //= p
	return n == 1
n = lmax + 1
while not smooth(n, (2, 3, 5, 7, 11)): n += 1
Nd = 2*n
c = [g for g in range(24, 641) if N % g == 0 and Nd % g == 0 and smooth(g, (2, 3, 5, 7)) and smooth(N//g, (2, 3, 5, 7)) and smooth(Nd//g, (2, 3, 5, 7))]
if not c: c = [g for g in range(48, 513) if N % g == 0 and smooth(g, (2, 3, 5, 7)) and smooth(N//g, (2, 3, 5, 7))]      # (no modulus realises ducc0's N_cc: the planner's own N_cc, any modulus of N)
print(" ".join(str(g) for g in c))
PY
)
	echo "== $cfg: candidates $gs" | tee -a $O/sweep_${TAG:-a}.txt
	for g in 0 $gs; do
		r=$(PXS_CHAIN_VERBOSE=1 PXS_THETA_G=$g timeout 120 python tools/chain_lab.py $cfg 3 2> $O/err.txt | tail -1)
		ch=$(grep -m1 "theta chain" $O/err.txt | cut -c1-160)
		echo "$cfg g=$g $r | $ch" | tee -a $O/sweep_${TAG:-a}.txt
	done
done
TAG=${TAG:-a} python tools/r06_planner_sweep_sum.py
