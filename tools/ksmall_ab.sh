#!/bin/bash
# rings per lane (K) of the VALU Legendre kernels at small sizes (lab build tools/libpxsht_klab.so): more, shorter waves fill the chip better when nm x nwave is small
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ksmall}; mkdir -p $O; cd $R; L=$R/tools/libpxsht_klab.so
for v in "PXS_X=0" "PXS_K_SMALL_OFF=1" "PXS_K_SMALL_OFF=1 PXS_K_SYN0=2 PXS_K_ANA0=2 PXS_K_SYNS=2 PXS_K_ANAS=2" "PXS_K_SMALL_OFF=1 PXS_K_SYN0=4 PXS_K_ANA0=4 PXS_K_SYNS=2 PXS_K_ANAS=3" "PXS_K_SMALL_OFF=1 PXS_K_SYN0=4 PXS_K_ANA0=4 PXS_K_SYNS=3 PXS_K_ANAS=4" "PXS_K_SMALL_OFF=1 PXS_K_SYN0=8 PXS_K_ANA0=8 PXS_K_SYNS=4 PXS_K_ANAS=4"; do
  env $v PIXELL_AMD_LIB=$L timeout 300 python tools/ksize_probe.py "[$v]" 2>&1 | grep lmax | tee -a $O/ksmall.txt
done
