#!/bin/bash
# tools/build_sched_variant.sh UNIT STRATEGY : build/lib_v_UNIT_STRATEGY.so = the product library with translation unit UNIT compiled
# under -amdgpu-sched-strategy=STRATEGY (A/B with tools/ab_libs.sh).  Re-uses the product build's other objects.
set -e
u=$1; st=$2
[ -n "$u" ] && [ -n "$st" ] || { echo "usage: $0 UNIT STRATEGY"; exit 1; }
cd "$(dirname "$0")/../dspsr_amd/csrc"
v=${u}_$st
o=../../build/obj_v_$v
mkdir -p "$o"
cp -p ../../build/obj/*.o "$o/"
rm -f "$o/$u.o" "$o/context.o"
make --no-print-directory -j2 OBJDIR=$o OUT=../../build/lib_v_$v.so BBENCH= "SCHED_$u=-mllvm -amdgpu-sched-strategy=$st" ../../build/lib_v_$v.so > "$o/log" 2>&1 || { tail -5 "$o/log"; exit 1; }
echo built build/lib_v_$v.so
