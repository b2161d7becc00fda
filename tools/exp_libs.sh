cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for round in 1 2; do
for n in "$@"; do
  export DSPSR_AMD_LIB=$GRAFT_REPO_ROOT/build/lib_$n.so
  rm -rf gpurun_out/el
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/el -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companions > gpurun_out/el.log 2>&1
  echo "== $n: $(python tools/kstats.py gpurun_out/el | grep -E "true" ) $(grep -o '"value": [0-9.]*' gpurun_out/el.log)"
done
done
