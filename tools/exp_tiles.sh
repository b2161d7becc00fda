# Tile-size / occupancy / launch-group experiments (timing only)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {  # name, extra bench args
  n=$1; shift
  rm -rf gpurun_out/e_$n
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e_$n -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/e_$n.log 2>&1
  echo "== $n $@ (LOG_POINTS=$DSPSR_AMD_LOG_POINTS WG_PER_CU=$DSPSR_AMD_WG_PER_CU)"; python tools/kstats.py gpurun_out/e_$n
}
export DSPSR_AMD_LOG_POINTS=13 DSPSR_AMD_WG_PER_CU=2; run p13w2 --max-parts 8
unset DSPSR_AMD_LOG_POINTS DSPSR_AMD_WG_PER_CU
run mp1 --max-parts 1
run mp2 --max-parts 2
run mp16 --max-parts 16
