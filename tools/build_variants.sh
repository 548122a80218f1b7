# builds libpxsht variants with different composite-radix sets compiled in -> variants/libpxsht_r<MAXR>.so
set -e
cd "$(dirname "$0")/.."; mkdir -p variants/obj
for r in "$@"; do
  for f in fftchain fft; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPXS_COMP_MAXR=$r -c pixell_amd/csrc/$f.hip -o variants/obj/$f.$r.o & done; wait
  objs=$(ls pixell_amd/build/*.o | grep -v -E '/(fftchain|fft)\.hip\.o'); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libpxsht_r$r.so $objs variants/obj/fftchain.$r.o variants/obj/fft.$r.o
done
ls -la variants/*.so
