# A/B of chain kernel variants on the GPU box
O=gpurun_out/chain_exp2; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(grep -E 'round-trip rms' $O/$tag.err | sed 's/.*round-trip/rt/') $(tail -1 $O/$tag.err)"; }
run nt512_1s PIXELL_AMD_LANES=0
cp pixell_amd/libpxsht.so /tmp/keep.so
for v in 768 1024; do cp gpurun_in/libpxsht_nt$v.so pixell_amd/libpxsht.so; run nt${v}_1s PIXELL_AMD_LANES=0; done
cp /tmp/keep.so pixell_amd/libpxsht.so
