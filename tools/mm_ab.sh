#!/bin/bash
# same-box A/B of the batched-analysis kernel forms (PXS_ANA_MM_FORM, lab switch) on the C4 bench; usage: mm_ab.sh <tag> "<forms>"
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; O=$R/gpurun_out/mmab_$TAG; mkdir -p $O; cd $R
for f in ${2:-0 1 2 3}; do
  PXS_ANA_MM_FORM=$f timeout 300 python bench.py --config c4 --no-cpu --steps 3 > $O/c4_form$f.json 2> $O/c4_form$f.err
  echo "form $f: $(grep 'stage ms' $O/c4_form$f.err) $(grep -o 'round-trip rms error [0-9.e-]*' $O/c4_form$f.err)"
done
PXS_ANA_MM_FORM=${3:-1} timeout 600 python -m pytest tests -m gpu -x -q -k "mm_analysis or batched or config4" 2>&1 | tail -3
