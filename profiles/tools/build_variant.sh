#!/bin/bash
# Builds an A/B variant of libsendslam_orb.so with extra compiler flags: build_variant.sh NAME [-DFLAG=V ...]
# -> send-slam_amd/lib/libexp_NAME.so (git-ignored); run with SENDSLAM_LIB=<that path>.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
NAME="$1"; shift
cd "$R/send-slam_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -shared \
    -o "lib/libexp_$NAME.so" csrc/ss_kernels.hip csrc/ss_api.cpp csrc/ss_geometry.cpp csrc/ss_track.cpp csrc/ss_pipe.cpp csrc/ss_xchg.hip
echo "$R/send-slam_amd/lib/libexp_$NAME.so"
