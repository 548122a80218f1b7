# A/B of radix sets compiled into the chain kernels (variants/libpxsht_r<MAXR>.so from tools/build_variants.sh)
O=gpurun_out/chain_exp3; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(tail -1 $O/$tag.err)"; }
V=$PWD/variants
run c3_r5 c3 PIXELL_AMD_LIB=$V/libpxsht_r5.so PXS_FFT_COMP=0
run c3_r16_plain c3 PXS_FFT_COMP=0
run c3_r8 c3 PIXELL_AMD_LIB=$V/libpxsht_r8.so PXS_FFT_RADICES=8,6,5,4,3,2
run c3_r10 c3 PIXELL_AMD_LIB=$V/libpxsht_r10.so PXS_FFT_RADICES=10,9,8,6,5,4,3,2
run c3_r10b c3 PIXELL_AMD_LIB=$V/libpxsht_r10.so PXS_FFT_RADICES=9,8,6,5,4,3,2
run c3_r5_again c3 PIXELL_AMD_LIB=$V/libpxsht_r5.so PXS_FFT_COMP=0
run c2_r5 c2 PIXELL_AMD_LIB=$V/libpxsht_r5.so PXS_FFT_COMP=0
run c2_r8 c2 PIXELL_AMD_LIB=$V/libpxsht_r8.so PXS_FFT_RADICES=8,6,5,4,3,2
run c2_r10 c2 PIXELL_AMD_LIB=$V/libpxsht_r10.so PXS_FFT_RADICES=10,9,8,6,5,4,3,2
