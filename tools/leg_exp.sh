# Legendre analysis reduction experiments
O=gpurun_out/leg_exp; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 2 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(grep -o 'round-trip rms error [0-9.e-]*' $O/$tag.err) $(tail -1 $O/$tag.err)"; }
V=$PWD/variants
run atomic c3 A=1
run nowait c3 PIXELL_AMD_LIB=$V/libpxsht_nowait.so
PIXELL_AMD_LIB=$V/libpxsht_nowait.so timeout 600 python -m pytest tests/test_sht_parity.py tests/test_baseline_configs.py -m gpu -x -q 2>&1 | tail -3
