#!/bin/bash
# same-box A/B of the C3 step: round 4's library (tools/libpxsht_r04.so, built from commit ce0499d) against the current one; usage: strip_ab.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05c}; mkdir -p $O; cd $R
for rep in 1 2; do for v in r04 cur; do
  L=$R/pixell_amd/libpxsht.so; [ $v = r04 ] && L=$R/tools/libpxsht_r04.so
  echo "$v: $(PIXELL_AMD_LIB=$L PXS_BENCH_NO_WEIGHTS=1 timeout 600 python bench.py --no-cpu --no-legs --steps 5 --warmup 2 2>&1 | grep -E "stage ms" | tail -1)" | tee -a $O/strip_ab.txt
done; done
