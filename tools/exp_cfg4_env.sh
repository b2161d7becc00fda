#!/bin/bash
# tools/exp_cfg4_env.sh TAG "ENV=.." "ENV2=.." ... : the cfg4 shard with an experiment build (build/lib_exp.so, DSPSR_AMD_LIB)
# under each environment string, alternating twice; value + per-kernel averages of one profiled run each
T=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
export DSPSR_AMD_LIB=$R/build/lib_exp.so
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e python3 bench.py --workload cfg4 --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/$T/v${i}_$rep.json 2> gpurun_out/$T/v${i}_$rep.err || { echo "failed [$e]"; tail -3 gpurun_out/$T/v${i}_$rep.err; exit 1; }
    echo "[$e] rep $rep: $(grep -o '"value": [0-9.]*' gpurun_out/$T/v${i}_$rep.json | head -1) $(grep -o '"status": "[a-z]*"' gpurun_out/$T/v${i}_$rep.json | head -1)"
  done
done
i=0
for e in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/$T/prof_$i
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof_$i -- python3 bench.py --workload cfg4 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/$T/prof_$i.log 2>&1
  echo "== [$e]"; python3 tools/kstats.py gpurun_out/$T/prof_$i | head -5
done
