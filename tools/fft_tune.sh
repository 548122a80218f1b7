cd $GRAFT_REPO_ROOT
# (the switches this script sets are read by LAB builds only: tools/build_variants.sh lab "-DPXS_LAB" <all stems>, then PIXELL_AMD_LIB=variants/libpxsht_lab.so)
for p in 512 1024 2048 4096 8192; do echo "== PXS_FFT_PTS=$p"; PXS_FFT_PTS=$p python tools/fft_bench.py 2>/dev/null | grep -E "\(77672, 216\)|\(262144, 64\)|\(16384, 1024\)|\(3106, 10800\)|\(776, 43200\)"; done
