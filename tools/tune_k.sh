cd $GRAFT_REPO_ROOT
# (the switches this script sets are read by LAB builds only: tools/build_variants.sh lab "-DPXS_LAB" <all stems>, then PIXELL_AMD_LIB=variants/libpxsht_lab.so)
run() { echo "== $*"; env "$@" python bench.py --config c2 --no-cpu 2>&1 >/dev/null | grep "stage ms"; }
run PXS_K_SYN0=4 PXS_K_ANA0=4 PXS_K_SYNS=2 PXS_K_ANAS=2
run PXS_K_SYN0=8 PXS_K_ANA0=8 PXS_K_SYNS=3 PXS_K_ANAS=3
run PXS_K_SYN0=8 PXS_K_ANA0=8 PXS_K_SYNS=4 PXS_K_ANAS=4
