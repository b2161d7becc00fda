#!/bin/bash
# tools/exp_cfg4_libs.sh TAG LIB1 LIB2 ... : the cfg4 shard with each library ("shipped" = dspsr_amd/libdspsr_amd.so, else a path
# for DSPSR_AMD_LIB), alternating twice, then per-kernel averages of one profiled run each
T=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
run() { if [ "$1" = shipped ]; then unset DSPSR_AMD_LIB; else export DSPSR_AMD_LIB=$R/$1; fi; shift; "$@"; }
for rep in 1 2; do
  for l in "$@"; do
    n=$(basename $l .so)
    run $l python3 bench.py --workload cfg4 --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/$T/${n}_$rep.json 2> gpurun_out/$T/${n}_$rep.err || { echo "failed [$l]"; tail -3 gpurun_out/$T/${n}_$rep.err; exit 1; }
    echo "[$l] rep $rep: $(grep -o '"value": [0-9.]*' gpurun_out/$T/${n}_$rep.json | head -1) $(grep -o '"status": "[a-z]*"' gpurun_out/$T/${n}_$rep.json | head -1)"
  done
done
for l in "$@"; do
  n=$(basename $l .so)
  rm -rf gpurun_out/$T/prof_$n
  run $l rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof_$n -- python3 bench.py --workload cfg4 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/$T/prof_$n.log 2>&1
  echo "== [$l]"; python3 tools/kstats.py gpurun_out/$T/prof_$n | head -5
done
