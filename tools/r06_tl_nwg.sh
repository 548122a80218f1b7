# phase clocks of a to_cc line (last wave of workgroup 0) with all 256 workgroups against a few: what part of the load phase is the other CUs' traffic
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_tl_nwg; mkdir -p $O
for n in 256 128 64 32 8; do
	echo "== PXS_TL_NWG=$n"; PXS_TL_NWG=$n PIXELL_AMD_LIB=tools/libpxsht_tl_time.so timeout 300 python tools/chain_lab.py c4 2 2>&1 | grep "lab\]" | tail -3 | cut -c1-400
done | tee $O/tl_nwg.txt
