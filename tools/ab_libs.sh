#!/bin/bash
# tools/ab_libs.sh TAG WORKLOAD ROUNDS LIB...  : same-box alternating A/B of library builds on one bench workload ("-" = the shipped
# library); prints Msamples/s and the roofline fraction per run, writes gpurun_out/TAG/ab_WORKLOAD.txt
T=$1; W=$2; N=$3; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
out=$R/gpurun_out/$T/ab_$W.txt
: > $out
for r in $(seq 1 $N); do
  for l in "$@"; do
    if [ "$l" = "-" ]; then unset DSPSR_AMD_LIB; else export DSPSR_AMD_LIB=$R/$l; fi
    a="--workload $W"; [ $W = target ] && a="--no-companions --no-h2d"
    v=$(timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline $a 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    echo "round $r lib $l $v" | tee -a $out
  done
done
