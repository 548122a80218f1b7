cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --config c3 --no-cpu --steps 2 2>&1 >/dev/null | grep "stage ms"; }
run PXS_PART_GB=1
run PXS_PART_GB=8
run PXS_K_SYN0=4
run PXS_K_ANAS=4
run PXS_K_ANAS=5
run PXS_K_ANAS=6
run PXS_K_ANA0=12
