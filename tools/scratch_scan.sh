#!/bin/bash
# tools/scratch_scan.sh [CSRC_DIR] : kernels of the library that use scratch memory (register spills or private arrays), per translation
# unit, compiled with the Makefile's flags for the unit (-Rpass-analysis=kernel-resource-usage).  A hot kernel that picks up spills
# after an unrelated change is a silent regression (round 5: the fused inverse pass, 28 bytes per lane, -15 %): run this after
# touching wgfft.h or a pass.
D=${1:-$(dirname "$0")/../dspsr_amd/csrc}
cd "$D"
for u in tfp fold fb_plain fb_conv1 fb_fwd_cols fb_fwd_rows fb_inv_chan fb_inv_chan_fold fb_inv_chan_search fb_two_pass fb_four_pass detect rescale scrunch sample_delay comm context; do
  [ -f $u.hip ] || continue
  s=$(grep "^SCHED_$u " Makefile | sed 's/^[^=]*= *//')
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $s -Rpass-analysis=kernel-resource-usage -c -o /tmp/scan_$$_$u.o $u.hip 2>&1 \
    | grep -E "Function Name|ScratchSize|    VGPRs:" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//; s/^.*remark: *//' | paste - - - \
    | awk -F'\t' -v u=$u '{ split($3, a, ": "); split($2, b, ": "); split($1, c, ": "); if (a[2] + 0 > 0) printf "%-20s scratch %4d B/lane  VGPRs %3d  %s\n", u, a[2], b[2], c[2] }'; rm -f /tmp/scan_$$_$u.o ) &
done
wait
