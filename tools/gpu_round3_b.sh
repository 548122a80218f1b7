#!/bin/bash
# chain-kernel lab baselines (per stage group, C3 and C4 shapes), skeleton variants, f64 MFMA lane-layout probe
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
./tools/mfma_probe.bin > $O/mfma_probe.txt 2>&1; head -3 $O/mfma_probe.txt
for cfg in c3 c4; do
  python tools/chain_lab.py $cfg 3 > $O/lab_${cfg}_base.json 2>> $O/lab.err; cat $O/lab_${cfg}_base.json
  PXS_CH_NOFFT=1 python tools/chain_lab.py $cfg 3 > $O/lab_${cfg}_nofft.json 2>> $O/lab.err; cat $O/lab_${cfg}_nofft.json
  PXS_CH_NOTW=1 python tools/chain_lab.py $cfg 3 > $O/lab_${cfg}_notw.json 2>> $O/lab.err; cat $O/lab_${cfg}_notw.json
  PXS_CH_NOFFT=1 PXS_CH_NOTW=1 python tools/chain_lab.py $cfg 3 > $O/lab_${cfg}_skel.json 2>> $O/lab.err; cat $O/lab_${cfg}_skel.json
done
rocprofv3 --kernel-trace --stats -d $O/prof_lab_c3 -o lab -- python tools/chain_lab.py c3 2 > /dev/null 2>> $O/lab.err
find $O/prof_lab_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/lab_c3_kernel_stats.csv
head -30 $O/lab_c3_kernel_stats.csv | cut -c1-200
PXS_BENCH_NREAL=10 timeout 600 python bench.py --no-cpu --config c5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -2 $O/bench_c5.err
python -c "import json; d=json.load(open('$O/bench_c5.json')); print(d['ms_per_realisation'], d['stage_ms_per_realisation'])"
timeout 600 python bench.py --no-cpu --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.err
