#!/bin/bash
# same-box A/B of the rings-per-lane (K) choice of the VALU analysis kernels against waves per SIMD (lab build: tools/libpxsht_klab.so, -DPXS_LAB);
# a wave issues one v_fma_f64 per 16 cycles, so a SIMD needs 4 resident waves for its FP64 rate (tools/dp_rate.hip)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-kwave}; mkdir -p $O; cd $R
./tools/dp_rate.bin | tee $O/dp_rate.txt
L=$R/tools/libpxsht_klab.so
for rep in 1 2; do for v in "PXS_K_ANAS=4" "PXS_K_ANAS=3" "PXS_K_ANAS=2" "PXS_K_ANA0=8" "PXS_K_ANA0=6" "PXS_K_ANA0=4"; do
  echo "$v: $(env $v PIXELL_AMD_LIB=$L PXS_BENCH_NO_WEIGHTS=1 timeout 600 python bench.py --no-cpu --no-legs --steps 4 --warmup 1 2>&1 | grep -E "stage ms" | tail -1)" | tee -a $O/kwave_ab.txt
done; done
