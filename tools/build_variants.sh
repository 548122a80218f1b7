# builds a variant of libpxsht.so for same-box kernel A/B runs (selected with PIXELL_AMD_LIB=variants/libpxsht_<name>.so)
# lab builds: pass -DPXS_LAB (plus -DPXS_LAB_MMTIME / -DPXS_LAB_NOATOM / -DPXS_LAB_NOLDSADD ...) for the switches that turn parts of a transform off (PXS_CH_NOFFT / NOTW / NOPH, PXS_FFT_DEBUG_NOPASS: wrong results, timing only); the default build does not contain them
# usage: tools/build_variants.sh <name> "<extra hipcc flags>" <source stems to recompile...>     e.g.  r10 "-DPXS_COMP_MAXR=10" fftchain fft
# (variants/ is in .gpurunignore: copy the .so under tools/ for the GPU call -- *.so is git-ignored -- as tools/fft2_ab.sh and tools/gpu_ntlab.sh expect, and delete it afterwards)
set -e
cd "$(dirname "$0")/.."; mkdir -p variants/obj
name=$1; flags=$2; shift; shift
ex=""
for f in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c pixell_amd/csrc/$f.hip -o variants/obj/$f.$name.o & ex="$ex|$f"; done; wait
objs=$(ls pixell_amd/build/*.o | grep -v -E "/(${ex#|})\.hip\.o")
vobjs=""; for f in "$@"; do vobjs="$vobjs variants/obj/$f.$name.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libpxsht_$name.so $objs $vobjs
ls -la variants/libpxsht_$name.so
