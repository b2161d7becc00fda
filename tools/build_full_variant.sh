#!/bin/bash
# tools/build_full_variant.sh NAME [-DFLAG ...] : experiment build of the WHOLE library (all geometries, the Makefile's
# per-pass scheduling flags) with -DDSPSR_AMD_EXPERIMENT -> build/lib_NAME.so; select it with DSPSR_AMD_LIB
set -e
cd "$(dirname "$0")/../dspsr_amd/csrc"
n=$1; shift
make --no-print-directory -j7 OBJDIR=../../build/obj_$n OUT=../../build/lib_$n.so \
  CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-result -Wno-unused-function -DDSPSR_AMD_EXPERIMENT $*" ../../build/lib_$n.so
echo built build/lib_$n.so
