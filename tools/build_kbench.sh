#!/bin/bash
# Build tools/kbench (production kernels) and tools/kbench_stamps (-DNODE_STAMPS diagnostic build).
set -e
cd "$(dirname "$0")/.."
SRC="neural-ode-features_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS tools/kbench.hip $SRC/kernels_layout.hip $SRC/kernels_pointwise.hip $SRC/kernels_conv.hip $SRC/kernels_wgrad.hip $SRC/kernels_head.hip $SRC/kernels_optim.hip $SRC/kernels_w4.hip $SRC/node_api.hip -o tools/kbench &
/opt/rocm/bin/hipcc $FLAGS -DNODE_STAMPS tools/kbench.hip $SRC/kernels_layout.hip $SRC/kernels_pointwise.hip $SRC/kernels_conv.hip $SRC/kernels_wgrad.hip $SRC/kernels_head.hip $SRC/kernels_optim.hip $SRC/kernels_w4.hip $SRC/node_api.hip -o tools/kbench_stamps &
wait
ls -la tools/kbench tools/kbench_stamps
