# tools/exp_p23.sh LIB WORKLOAD "ENV=.." ... : bench value per environment string (experiment build build/lib_LIB.so), twice, alternating
cd $GRAFT_REPO_ROOT
export DSPSR_AMD_LIB=$GRAFT_REPO_ROOT/build/lib_$1.so
w=$2; shift; shift
a="--workload $w"; [ $w = target ] && a="--no-companions"
for r in 1 2; do
for e in "$@"; do
  echo "== $w [$e] $(env $e python bench.py $a --no-cpu-baseline --steps 30 --warmup 4 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done; done
