# round 6: where the time of the single-kernel theta engine goes -- lab variants of thetaline.hip with parts switched off (wrong
# results, timing only), each timed with tools/chain_lab.py on the C4 configuration.  Run on the GPU box: bash tools/tl_lab.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r06_tl_lab}; mkdir -p $O
for v in base nocomp noxchg noboth; do
	f=tools/libpxsht_tl_$v.so
	[ -f $f ] || continue
	PIXELL_AMD_LIB=$f python tools/chain_lab.py c4 5 2>&1 | tail -1 | sed "s/^/$v /"
done | tee $O/tl_lab_c4.txt
[ -f tools/libpxsht_tl_time.so ] && PIXELL_AMD_LIB=tools/libpxsht_tl_time.so python tools/chain_lab.py c4 2 2>&1 | grep "lab\]" | tail -2 | tee $O/tl_phase_times_c4.txt
