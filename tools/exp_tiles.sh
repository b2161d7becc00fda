cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "14 1" "13 2"; do
  set -- $cfg
  export DSPSR_AMD_LOG_POINTS=$1 DSPSR_AMD_WG_PER_CU=$2
  rm -rf gpurun_out/e_$1
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e_$1 -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "logP=$1 wg/cu=$2"; python tools/kstats.py gpurun_out/e_$1
done
