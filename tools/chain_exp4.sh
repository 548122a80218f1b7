# how the chain kernels respond to workgroups per CU (extra dynamic LDS lowers it)
O=gpurun_out/chain_exp4; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 2 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(tail -1 $O/$tag.err)"; }
run pad0 c3 A=1
run pad12k c3 PXS_CH_LDS_PAD=12288
run pad35k c3 PXS_CH_LDS_PAD=35840
run pad80k c3 PXS_CH_LDS_PAD=81920
