# HBM traffic per kernel from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes)
# usage: bash tools/pmc.sh [max_parts]   -> prints mean KB per dispatch and kernel
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MP=${1:-8}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --max-parts $MP > /dev/null 2>&1
  f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv,sys,collections
f,c=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r.get('Kernel_Name','')
    if 'dspsr' not in n: continue
    n=n.split('(')[0].replace('void dspsr_amd::','')
    acc[n].append(float(r['Counter_Value']))
for n,v in acc.items():
    print("%s %s: n=%d mean=%.1f KB/dispatch"%(c,n,len(v),sum(v)/len(v)))
PY
done
