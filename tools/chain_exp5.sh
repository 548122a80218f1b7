# the shared modulus g of the theta chains (PXS_THETA_G): candidates with the same N_cc and M
O=gpurun_out/chain_exp5; mkdir -p $O
run() { tag=$1; cfg=$2; shift; shift; env PXS_CHAIN_VERBOSE=1 "$@" timeout 300 python bench.py --config $cfg --no-cpu --steps 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(grep -m1 'theta chain' $O/$tag.err | cut -c1-150)"; echo "   $(tail -1 $O/$tag.err)"; }
run c3_g225 c3 A=1
run c3_g270 c3 PXS_THETA_G=270
run c3_g150 c3 PXS_THETA_G=150
run c3_g135 c3 PXS_THETA_G=135
run c2_g60 c2 A=1
run c2_g100 c2 PXS_THETA_G=100
run c2_g150 c2 PXS_THETA_G=150
run c2_g90 c2 PXS_THETA_G=90
run c2_g75 c2 PXS_THETA_G=75
