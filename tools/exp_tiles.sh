cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "14 1" "13 2" "12 4"; do
  set -- $cfg
  export DSPSR_AMD_LOG_POINTS=$1 DSPSR_AMD_WG_PER_CU=$2
  rm -rf gpurun_out/e_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e_$1 -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --max-parts 4 > /dev/null 2>&1
  f=$(find gpurun_out/e_$1 -name "*kernel_stats.csv")
  echo "logP=$1 wg/cu=$2: $(grep dspsr $f | sed 's/(dspsr_amd::FbGeom[^"]*"/"/; s/(float const[^"]*"/"/' | awk -F, '{printf "%s %.0f | ", $1, $4/1000}')"
done
