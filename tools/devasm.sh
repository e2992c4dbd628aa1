#!/bin/bash
# Diagnostic (build container, no GPU): device assembly of core.hip with the product's flags -> build_ab/core.s, then
# registers / scratch of the kernels whose mangled name matches $1 (e.g. pf_score_kernelILb0ELi16).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -mfma \
      -mllvm -disable-machine-licm $EXTRA --cuda-device-only -S -o build_ab/core.s laser_slam_amd/csrc/core.hip 2> build_ab/core.err || { grep -A3 "error:" build_ab/core.err; exit 1; }
[ -n "$1" ] && grep -E "^\s*\.set .*$1.*\.(num_vgpr|numbered_sgpr|private_seg_size)," build_ab/core.s | sed 's/^\s*\.set //'
exit 0
