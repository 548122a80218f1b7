cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/healpix; mkdir -p $O
cd $R; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hp -o hp -- python tools/healpix_bench.py > $O/prof.log 2>&1
f=$(find /tmp/prof_hp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -14 "$f" | cut -c1-160
tail -3 $O/prof.log
