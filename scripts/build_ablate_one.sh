#!/bin/bash
# rebuild ONE object of the -DMSCKF_ABLATE library and relink (the full `make ablate` compiles the eight files one after the other)
#   scripts/build_ablate_one.sh kernels_feature
set -e
cd "$(dirname "$0")/../msckf_mono_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-variable -Wno-unused-result -ffp-contract=fast -fno-slp-vectorize -DMSCKF_ABLATE"
for f in "$@"; do /opt/rocm/bin/hipcc $F -c $f.hip -o ../lib_ab/$f.o; done
O="kernels_state kernels_feature kernels_qr kernels_gram kernels_chol kernels_kalman kernels_literal msckf_hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib_ab/libmsckf_hip_ablate.so $(for o in $O; do echo ../lib_ab/$o.o; done) -ldl
