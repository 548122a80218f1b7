# ring stages at 8 waves per SIMD: tile size so that 4 workgroups fit the LDS, or the full 2560-point tiles
O=gpurun_out/chain_exp4; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 2 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(tail -1 $O/$tag.err)"; }
run tile40 c3 A=1
run tilefull c3 PXS_RING_TILE_KB=0
run tile52 c3 PXS_RING_TILE_KB=52
run tile40_c4 c4 A=1
run tilefull_c4 c4 PXS_RING_TILE_KB=0
run tile40_c2 c2 A=1
run tilefull_c2 c2 PXS_RING_TILE_KB=0
