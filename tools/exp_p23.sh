# tools/exp_p23.sh : pass 2 + inverse in sub-groups (default) against whole launches, full experiment build (build/lib_full.so)
cd $GRAFT_REPO_ROOT
export DSPSR_AMD_LIB=$GRAFT_REPO_ROOT/build/lib_full.so
for w in cfg2; do
for r in 1 2; do
for e in "X=1" "DSPSR_AMD_P23_SUB=0" "DSPSR_AMD_P23_SUB=64"; do
  echo "== $w [$e] $(env $e python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done; done; done
