#!/bin/bash
# chain_lab.py under a list of environment settings: LAB_ENVS="A=1 B=2;C=3" (semicolon-separated sets), LAB_CFGS, LAB_OUT
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${LAB_OUT:-labenv}; mkdir -p $O
IFS=';' read -ra SETS <<< "$LAB_ENVS"
for cfg in ${LAB_CFGS:-c3}; do
  for set in "" "${SETS[@]}"; do
    echo "== $cfg [$set]" | tee -a $O/lab.txt
    env $set python tools/chain_lab.py $cfg 3 2>> $O/lab.err | tee -a $O/lab.txt
  done
done
