#!/bin/bash
# runs tools/chain_lab.py for the stock library and every variants/libpxsht_*.so named on the command line (or all)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${LAB_OUT:-lab}; mkdir -p $O
cfgs=${LAB_CFGS:-"c3 c4"}
libs="$@"; [ -z "$libs" ] && libs=$(ls variants/libpxsht_*.so 2>/dev/null)
for cfg in $cfgs; do
  python tools/chain_lab.py $cfg 3 2>> $O/lab.err | tee -a $O/lab_$cfg.jsonl
  for l in $libs; do PIXELL_AMD_LIB=$PWD/$l python tools/chain_lab.py $cfg 3 2>> $O/lab.err | tee -a $O/lab_$cfg.jsonl; done
done
