#!/bin/bash
# A/B of the second-generation chain kernel per stage (PXS_CHAIN_V2 bit masks) on tools/chain_lab.py; usage: gpu_v2lab.sh <out tag> [cfgs] [libs...]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; ulimit -c 0
O=gpurun_out/${1:-v2lab}; mkdir -p $O; CFGS=${2:-c3}; shift; shift
MASKS=${V2_MASKS:-"0 2 4 8 16 64 256 512 1 32 128 0x35e -1"}
for cfg in $CFGS; do
  for lib in "" "$@"; do
    for m in $MASKS; do
      echo "== $cfg lib=[$lib] mask=$m" | tee -a $O/lab.txt
      ( [ -n "$lib" ] && export PIXELL_AMD_LIB=$PWD/$lib; PXS_CHAIN_V2=$m timeout 300 python tools/chain_lab.py $cfg 3 2>> $O/lab.err | tee -a $O/lab.txt ) || echo "FAILED rc=$?" | tee -a $O/lab.txt
    done
  done
done
