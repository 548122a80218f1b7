# recurrence seeds on / off
O=gpurun_out/leg_exp; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(grep -o 'round-trip rms error [0-9.e-]*' $O/$tag.err) $(tail -1 $O/$tag.err)"; }
run seeds c3 A=1
run noseeds c3 PXS_SEED_GB=0
run seeds_c2 c2 A=1
run noseeds_c2 c2 PXS_SEED_GB=0
run seeds_c4 c4 A=1
run noseeds_c4 c4 PXS_SEED_GB=0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
