# sweeps of the planner's choices with tools/chain_lab.py: shared modulus of the theta chains (PXS_THETA_G), first factor of the ring FFT splits
# (the switches this script sets are read by LAB builds only: tools/build_variants.sh lab "-DPXS_LAB" <all stems>, then PIXELL_AMD_LIB=variants/libpxsht_lab.so)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-sweep}; mkdir -p $O; CFG=${2:-c3}
for g in ${GS:-0 96 120 144 160 180 192 240 288 320 480}; do
  echo "== $CFG PXS_THETA_G=$g" | tee -a $O/sweep.txt
  PXS_CHAIN_VERBOSE=1 PXS_THETA_G=$g timeout 200 python tools/chain_lab.py $CFG 3 2> $O/err.txt | tee -a $O/sweep.txt; grep "theta chain" $O/err.txt | sort -u | tee -a $O/sweep.txt
done
for a in ${AS_ANA:-135 144 150 160 180 200 216 225 240 270}; do
  echo "== $CFG PXS_RING_A_ANA=$a" | tee -a $O/sweep.txt
  PXS_RING_A_ANA=$a timeout 200 python tools/chain_lab.py $CFG 3 2>/dev/null | tee -a $O/sweep.txt
done
for a in ${AS_SYN:-200 216 240 270 288 300 320 360 400}; do
  echo "== $CFG PXS_RING_A_SYN=$a" | tee -a $O/sweep.txt
  PXS_RING_A_SYN=$a timeout 200 python tools/chain_lab.py $CFG 3 2>/dev/null | tee -a $O/sweep.txt
done
