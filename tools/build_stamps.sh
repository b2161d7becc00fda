#!/bin/bash
# tools/build_stamps.sh ID : diagnostic library build/lib_st<ID>.so with the phase stamps of kernel ID compiled in (csrc/stamps.h;
# ids: 1 k_fwd_cols, 2 k_fwd_rows, 3 k_inv_chan, 4 k_inv_a, 6 k_fwd_col1q, 7 k_rows_inv, 8 k_tfp).  Re-uses the objects of the
# product build for every other translation unit.  Read with: DSPSR_AMD_LIB=build/lib_st<ID>.so python tools/stamps.py <ID> [workload]
set -e
id=$1
cd "$(dirname "$0")/../dspsr_amd/csrc"
make --no-print-directory BBENCH= >/dev/null
case $id in
  1) u="fb_fwd_cols";; 2) u="fb_fwd_rows";; 3) u="fb_inv_chan fb_inv_chan_fold";; 4) u="fb_four_pass";; 6|7) u="fb_two_pass";; 8) u="tfp";;
  *) echo "unknown kernel id $id"; exit 1;;
esac
o=../../build/obj_st$id
rm -rf $o && mkdir -p $o && cp -p ../../build/obj/*.o $o/
for x in $u context; do rm -f $o/$x.o; done
make --no-print-directory -j8 OBJDIR=$o OUT=../../build/lib_st$id.so BBENCH= EXTRA=-DFB_STAMPS=$id ../../build/lib_st$id.so
echo built build/lib_st$id.so
