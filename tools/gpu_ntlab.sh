# per-stage A/B of the workgroup size of the chain kernels (variant builds with -DPXS_CH_NT2=<n>; PXS_CH_NT2_MASK: bit SID -> that stage runs with n threads)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-ntlab}; mkdir -p $O; CFG=${2:-c3}
for lib in tools/libpxsht_nt256.so tools/libpxsht_nt384.so; do
  [ -f $lib ] || continue
  for m in 0 0x1 0x2 0x4 0x8 0x10 0x20 0x40 0x80 0x100 0x1ff; do
    echo "== $CFG $lib mask=$m" | tee -a $O/lab.txt
    PIXELL_AMD_LIB=$PWD/$lib PXS_CH_NT2_MASK=$m timeout 200 python tools/chain_lab.py $CFG 3 2>> $O/lab.err | tee -a $O/lab.txt
  done
done
