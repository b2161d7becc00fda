# LDS bank-conflict share per kernel: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (one pass)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/pmc_lds -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n=r.get('Kernel_Name','')
    if 'dspsr' not in n: continue
    n=n.split('(')[0].replace('void dspsr_amd::','')
    acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,d in acc.items():
    c=sum(d['SQ_LDS_BANK_CONFLICT'])/max(1,len(d['SQ_LDS_BANK_CONFLICT'])); a=sum(d['SQ_LDS_IDX_ACTIVE'])/max(1,len(d['SQ_LDS_IDX_ACTIVE']))
    print("%-28s LDS active %.3e cycles/dispatch, bank conflict %.3e (%.1f %%)" % (n, a, c, 100*c/max(a,1)))
PY
