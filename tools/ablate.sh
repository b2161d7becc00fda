# Ablation of the filterbank passes (DSPSR_AMD_DEBUG bits: 1 no stores, 2 no loads, 8 no pass-1 twiddle;
# results are wrong when set) -- average kernel time per launch group from rocprofv3 --kernel-trace --stats.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MP=${MP:-8}
for d in ${DBG_LIST:-0 1 2 3 8}; do
  export DSPSR_AMD_DEBUG=$d
  rm -rf gpurun_out/abl_$d
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$d -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --max-parts $MP > /dev/null 2>&1
  echo "dbg=$d max_parts=$MP"; python tools/kstats.py gpurun_out/abl_$d
done
