#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : experiment build of the library -> build/lib_NAME.so (headline kernels only;
# -DDSPSR_AMD_EXPERIMENT: the only builds that read the DSPSR_AMD_* environment knobs and ablation bits)
set -e
cd "$(dirname "$0")/../dspsr_amd/csrc"
n=$1; shift
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-result -DFB_ONLY_HEADLINE -DDSPSR_AMD_EXPERIMENT "$@" -shared \
  -o ../../build/lib_$n.so context.hip filterbank.hip tfp.hip detect.hip fold.hip rescale.hip sample_delay.hip comm.hip host_prep.cpp -ldl
echo built build/lib_$n.so
