#!/bin/bash
# tools/build_flag_variant.sh NAME UNIT FLAGS... : build/lib_f_NAME.so = the product library with translation unit UNIT compiled with
# its Makefile flags PLUS the given extra flags (codegen lottery tickets: -mllvm knobs; A/B with tools/ab_libs.sh)
set -e
n=$1; u=$2; shift 2
cd "$(dirname "$0")/../dspsr_amd/csrc"
o=../../build/obj_f_$n
mkdir -p "$o"
cp -p ../../build/obj/*.o "$o/"
rm -f "$o/$u.o" "$o/context.o"
s=$(grep "^SCHED_$u " Makefile | sed 's/^[^=]*= *//')
make --no-print-directory -j2 OBJDIR=$o OUT=../../build/lib_f_$n.so BBENCH= "SCHED_$u=$s $*" ../../build/lib_f_$n.so > "$o/log" 2>&1 || { tail -5 "$o/log"; exit 1; }
echo built build/lib_f_$n.so
