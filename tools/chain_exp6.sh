# ring-FFT stages in passes whose intermediate fits the 256 MB memory-side cache (PXS_RING_CHUNK_MB)
O=gpurun_out/chain_exp6; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(tail -1 $O/$tag.err)"; }
run all c3 A=1
run mb64 c3 PXS_RING_CHUNK_MB=64
run mb128 c3 PXS_RING_CHUNK_MB=128
run mb200 c3 PXS_RING_CHUNK_MB=200
run mb512 c3 PXS_RING_CHUNK_MB=512
run mb2048 c3 PXS_RING_CHUNK_MB=2048
run c4_all c4 A=1
run c4_mb128 c4 PXS_RING_CHUNK_MB=128
