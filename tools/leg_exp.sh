O=gpurun_out/leg_exp; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(grep -o 'round-trip rms error [0-9.e-]*' $O/$tag.err) $(tail -1 $O/$tag.err)"; }
run c4_new c4 A=1
run c4_viacc c4 PXS_SYN_VIA_CC=1
run c2_new c2 A=1
run c2_viacc c2 PXS_SYN_VIA_CC=1
run c3_new c3 A=1
