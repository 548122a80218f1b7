#!/bin/bash
# C5 pipeline, same box: synthesis through the CC grid (default at 10800 rings / lmax 6000) against the direct Legendre synthesis on the map's rings
# (PXS_SYN_VIA_CC=0; lab build tools/libpxsht_shtlab.so, -DPXS_LAB)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-c5ab}; mkdir -p $O; cd $R
L=$R/tools/libpxsht_shtlab.so
for rep in 1 2; do for v in "PXS_X=0" "PXS_SYN_VIA_CC=0"; do
  echo "$v: $(env $v PIXELL_AMD_LIB=$L timeout 600 python bench.py --config c5 --no-cpu --no-legs --steps 1 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_realisation'], d['stage_ms_per_realisation'], d['roofline']['kernel_ms_per_realisation'])")" | tee -a $O/c5_ab.txt
done; done
