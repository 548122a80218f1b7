#!/bin/bash
# K retune of the Legendre kernels, cpu_baseline check, PMC traffic passes of the C3 bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
run() { echo "== $*" | tee -a $O/ktune.txt; env "$@" python bench.py --config ${CFG:-c3} --no-cpu --steps 2 2>&1 >/dev/null | grep "stage ms" | tee -a $O/ktune.txt; }
run PXS_K_SYNS=3
run PXS_K_SYNS=4
run PXS_K_SYN0=8
run PXS_K_ANAS=3
CFG=c4 run PXS_K_SYN0=4
CFG=c4 run PXS_K_SYN0=8
CFG=c4 run PXS_K_ANA0=4
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.err
python -c "import json; d=json.load(open('$O/bench_c3.json')); print(json.dumps(d['cpu_baseline'])[:1500])"
bash tools/pmc_traffic.sh c3 > $O/pmc.log 2>&1; tail -3 $O/pmc.log
python tools/pmc_traffic_sum.py c3 r03 > $O/traffic_c3.txt 2>&1; cp profiles/r03_traffic_c3.json $O/ 2>/dev/null; tail -25 $O/traffic_c3.txt | cut -c1-170
