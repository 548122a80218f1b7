#!/bin/bash
# builds and runs tools/dma_skel.hip on the GPU box; output -> gpurun_out/<tag>/dma_skeleton.txt (copied to profiles/ by hand)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-skel}; mkdir -p $O
for fl in 3 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DFL=$fl tools/dma_skel.hip -o /tmp/dma_skel.bin 2>> $O/skel.err && timeout 300 /tmp/dma_skel.bin | tee -a $O/dma_skeleton.txt
done
