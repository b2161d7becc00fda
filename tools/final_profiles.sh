#!/bin/bash
# tools/final_profiles.sh TAG : on the GPU box -- the committed evidence of a round from ONE box: rocprofv3 kernel stats + bench
# line of the default workload and of every other BASELINE workload (gpurun_out/TAG/<wl>_kernel_stats.txt, <wl>_bench.json), the
# PMC traffic of every workload (tools/pmc_traffic.sh -> gpurun_out/TAG/<wl>_traffic.json) and the driver-form default line
T=$1
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
WL=${WL:-target cfg1 cfg1opt cfg2 cfg3 cfg4 cfg5 cfg5c odd_nchan odd_fres after after8k after1k after8c plain fold}
PHASE=${PHASE:-all}      # stats | pmc | all (a call on the GPU box is limited to 20 minutes: run the two phases as two calls if need be)
# the counters first: the bench lines of the stats phase quote the traffic files of THIS build (bench.py refuses any other)
if [ $PHASE != stats ]; then
for w in $WL; do
  bash tools/pmc_traffic.sh $T $w > gpurun_out/$T/pmc_$w.txt 2>&1 || { echo "pmc $w failed"; tail -3 gpurun_out/$T/pmc_$w.txt; exit 1; }
  head -1 gpurun_out/$T/pmc_$w.txt | cut -c1-400
  rm -rf gpurun_out/$T/pmc_${w}_*_SIZE
  cp gpurun_out/$T/${w}_traffic.json profiles/${T}_${w}_traffic.json     # on the box: the stats phase's bench lines then carry this build's traffic
done
fi
if [ $PHASE != pmc ]; then
for w in $WL; do
  a="--workload $w"; [ $w = target ] && a="--no-companions --no-h2d"
  rm -rf gpurun_out/$T/prof_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof_$w -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $a > gpurun_out/$T/${w}_bench.json 2> gpurun_out/$T/${w}.err || { echo "$w failed"; tail -3 gpurun_out/$T/${w}.err; exit 1; }
  python3 tools/kstats.py gpurun_out/$T/prof_$w > gpurun_out/$T/${w}_kernel_stats.txt
  cp $(ls gpurun_out/$T/prof_$w/*/*_kernel_stats.csv | head -1) gpurun_out/$T/${w}_kernel_stats.csv
  rm -rf gpurun_out/$T/prof_$w
  echo "== $w: $(grep -o '"value": [0-9.]*' gpurun_out/$T/${w}_bench.json | head -1) $(grep -o '"frac": [0-9.]*' gpurun_out/$T/${w}_bench.json | head -1)"
  head -6 gpurun_out/$T/${w}_kernel_stats.txt
done
fi
