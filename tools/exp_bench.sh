# tools/exp_bench.sh "bench args" ... : plain bench.py runs (no profiler), one JSON value per argument string
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-companions $a > gpurun_out/eb.log 2>&1
  echo "== [$a]  $(grep -o '"value": [0-9.]*' gpurun_out/eb.log | head -1)  $(grep -o '"frac": [0-9.]*' gpurun_out/eb.log | tr '\n' ' ')"
done
