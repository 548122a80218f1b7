# is the Legendre stage waiting for its coefficient rows?
O=gpurun_out/leg_exp; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 2 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(tail -1 $O/$tag.err)"; }
V=$PWD/variants
run base c3 A=1
run coefhot c3 PIXELL_AMD_LIB=$V/libpxsht_coefhot.so PXS_BENCH_NOCHECK=1
run base_c2 c2 A=1
run coefhot_c2 c2 PIXELL_AMD_LIB=$V/libpxsht_coefhot.so PXS_BENCH_NOCHECK=1
