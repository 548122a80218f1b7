#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-kbig}; mkdir -p $O; cd $R; L=$R/tools/libpxsht_klab.so
for cfg in c2 c3; do for rep in 1 2; do for v in "PXS_X=0" "PXS_K_SYN0=2 PXS_K_SYNS=2" "PXS_K_SYN0=2" "PXS_K_SYNS=2" "PXS_K_ANA0=4"; do
  echo "$cfg [$v]: $(env $v PIXELL_AMD_LIB=$L PXS_BENCH_NO_WEIGHTS=1 timeout 600 python bench.py --config $cfg --no-cpu --no-legs --steps 5 --warmup 2 2>&1 | grep -E "stage ms" | tail -1)" | tee -a $O/kbig.txt
done; done; done
