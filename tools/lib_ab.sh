#!/bin/bash
# same-box A/B of library variants on a bench command; usage: lib_ab.sh <tag> "<variant names under tools/libpxsht_<name>.so; cur = the product library>" [bench args...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; cd $R; V=$2; shift; shift
ARGS=${@:---no-cpu --no-legs --steps 5 --warmup 2}
for rep in 1 2; do for v in $V; do
  L=$R/tools/libpxsht_$v.so; [ $v = cur ] && L=$R/pixell_amd/libpxsht.so
  echo "$v: $(PIXELL_AMD_LIB=$L PXS_BENCH_NO_WEIGHTS=1 timeout 600 python bench.py $ARGS 2>&1 | grep -E "stage ms" | tail -1)" | tee -a $O/lib_ab.txt
done; done
