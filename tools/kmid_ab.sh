#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-kmid}; mkdir -p $O; cd $R; L=$R/tools/libpxsht_klab.so
for v in "PXS_X=0" "PXS_K_SYN0=2 PXS_K_SYNS=2" "PXS_K_SYN0=8 PXS_K_SYNS=4" "PXS_K_ANA0=4 PXS_K_ANAS=3" "PXS_K_ANA0=4 PXS_K_ANAS=4" "PXS_K_ANA0=8 PXS_K_ANAS=3"; do
  env $v PIXELL_AMD_LIB=$L timeout 300 python tools/kmid_probe.py "[$v]" 2>&1 | grep lmax | tee -a $O/kmid.txt
done
