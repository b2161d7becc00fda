# tools/exp_wl.sh "bench args" ... : kernel stats + bench line for each argument string (main library)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for a in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/ew_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ew_$i -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline $a > gpurun_out/ew_$i.log 2>&1
  echo "== [$a]"; python tools/kstats.py gpurun_out/ew_$i; grep -o '"value": [0-9.]*' gpurun_out/ew_$i.log | head -1; grep -o '"frac": [0-9.]*' gpurun_out/ew_$i.log | head -1
done
