# same-box A/B of the 2-D FFT with and without the radix-7 code in the chain kernels (tools/libpxsht_nor7.so: tools/build_variants.sh nor7 "-DPXS_NO_RADIX7" fftchain)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/fft2ab; mkdir -p $O
for i in 1 2; do
  python tools/fft2_bench.py 2>/dev/null | tee -a $O/ab.txt
  PIXELL_AMD_LIB=$PWD/tools/libpxsht_nor7.so python tools/fft2_bench.py 2>/dev/null | tee -a $O/ab.txt
done
python tools/chain_lab.py c3 3 2>/dev/null | tee -a $O/ab.txt
PIXELL_AMD_LIB=$PWD/tools/libpxsht_nor7.so PXS_THETA_DUCC_NCC=0 PXS_ANALYSIS=interpolant python tools/chain_lab.py c3 3 2>/dev/null | tee -a $O/ab.txt
PXS_THETA_DUCC_NCC=0 PXS_ANALYSIS=interpolant python tools/chain_lab.py c3 3 2>/dev/null | tee -a $O/ab.txt
