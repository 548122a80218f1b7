# Run on the GPU box: tests, bench (default config), rocprof kernel stats of the same bench command, PMC traffic pass.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r01}
cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1; tail -3 $O/pytest_gpu_$TAG.log
python bench.py > $O/bench_c3_$TAG.json 2> $O/bench_c3_$TAG.err; tail -4 $O/bench_c3_$TAG.err; cat $O/bench_c3_$TAG.json
python bench.py --config c2 --no-cpu > $O/bench_c2_$TAG.json 2> $O/bench_c2_$TAG.err; tail -2 $O/bench_c2_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_$TAG -o c3 -- python $R/bench.py --no-cpu --steps 2 > $O/prof_c3_$TAG.log 2>&1
ls $O/prof_c3_$TAG
