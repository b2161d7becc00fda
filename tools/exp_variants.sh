# A/B of experiment builds: tools/exp_variants.sh name1 name2 ...  (build/lib_NAME.so), kernel stats per variant
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in "$@"; do
  export DSPSR_AMD_LIB=$GRAFT_REPO_ROOT/build/lib_$n.so
  rm -rf gpurun_out/v_$n
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v_$n -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-companions > gpurun_out/v_$n.log 2>&1
  echo "== $n"; python tools/kstats.py gpurun_out/v_$n; grep -o '"value": [0-9.]*' gpurun_out/v_$n.log
done
