#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
./tools/dma_skel.bin | tee $O/dma_skel.txt
for l in "" variants/libpxsht_redmfma.so; do
  PIXELL_AMD_LIB=${l:+$PWD/$l} timeout 600 python bench.py --no-cpu --steps 3 --warmup 1 2> $O/b.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['frac_hw_both'], d['roundtrip_rms_error'])" | tee -a $O/redmfma.txt
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 > $GRAFT_REPO_ROOT/$O/prof_a.log 2>&1
f=$(find /tmp/prof_a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/c3_kernel_stats.csv
PIXELL_AMD_LIB=$GRAFT_REPO_ROOT/variants/libpxsht_redmfma.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 > $GRAFT_REPO_ROOT/$O/prof_b.log 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/c3_kernel_stats_redmfma.csv
cd $GRAFT_REPO_ROOT; head -8 $O/c3_kernel_stats.csv | cut -c1-160; grep leg_ana $O/c3_kernel_stats_redmfma.csv | cut -c1-160
