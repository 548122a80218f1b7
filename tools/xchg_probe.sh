# builds and runs the exchange probes on the GPU box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r06_xchg}; mkdir -p $O
for v in "" "-DXP_SEQ=16,16,9,7"; do
	hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ipixell_amd/csrc $v tools/xchg_probe.hip -o /tmp/xp 2>/dev/null && echo "== workgroup-wide $v" && /tmp/xp
done 2>&1 | tee $O/xchg_probe.txt
for v in "-DXP_SEQ=8,2,9,7" "-DXP_SEQ=4,4,9,7" "-DXP_SEQ=16,9,7" "-DXP_SEQ=9,5,5,3" "-DXP_SEQ=8,9,7"; do
	hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ipixell_amd/csrc -mllvm -simplifycfg-sink-common=false $v tools/xchg_probe2.hip -o /tmp/xp2 2>/dev/null && echo "== wave-private $v" && /tmp/xp2
done 2>&1 | tee -a $O/xchg_probe.txt
