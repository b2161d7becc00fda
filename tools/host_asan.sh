#!/bin/bash
# Host-side preparation code (csrc/host_prep.cpp: chirp builder, smearing, FFT-length choice, bin plan, -K delays) under
# AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool).  usage: bash tools/host_asan.sh
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared -I include -I dspsr_amd/csrc \
    -o /tmp/libhp_asan.so dspsr_amd/csrc/host_prep.cpp
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python tools/host_asan_run.py
