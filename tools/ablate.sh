cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 8 5; do
  export DSPSR_AMD_DEBUG=$d
  rm -rf gpurun_out/abl_$d
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$d -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --max-parts 4 > /dev/null 2>&1
  f=$(find gpurun_out/abl_$d -name "*kernel_stats.csv")
  echo "dbg=$d: $(grep dspsr $f | sed 's/(dspsr_amd::FbGeom[^"]*"/"/; s/(float const[^"]*"/"/' | awk -F, '{printf "%s %.0f | ", $1, $4/1000}')"
done
