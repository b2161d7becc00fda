# tools/exp_env.sh LIBNAME "ENV=.. ENV2=.." ... : one kernel-stats run per environment string
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
n=$1; shift
export DSPSR_AMD_LIB=$GRAFT_REPO_ROOT/build/lib_$n.so
i=0
for e in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/ee_$i
  env $e timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ee_$i -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/ee_$i.log 2>&1
  echo "== $n [$e]"; python tools/kstats.py gpurun_out/ee_$i; grep -o '"value": [0-9.]*' gpurun_out/ee_$i.log
done
