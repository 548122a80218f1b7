# timing experiments of the fused FFT chains on the GPU box (results of the NOFFT / NOTW variants are wrong on purpose)
O=gpurun_out/chain_exp; mkdir -p $O
run() { tag=$1; shift; env PIXELL_AMD_LANES=0 PXS_BENCH_NOCHECK=1 "$@" timeout 300 python bench.py --no-cpu --steps 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(tail -1 $O/$tag.err)"; }
run base
run nofft PXS_CH_NOFFT=1
run notw PXS_CH_NOTW=1
run nofft_notw PXS_CH_NOFFT=1 PXS_CH_NOTW=1
