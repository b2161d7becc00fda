# SQ issue/wait counters per kernel (one pass, 8 SQ slots)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/pmc_sq.log 2>&1
f=$(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n=r.get('Kernel_Name','')
    if 'dspsr' not in n: continue
    n=n.split('(')[0].replace('void dspsr_amd::','')
    acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,d in acc.items():
    m={k:sum(v)/len(v) for k,v in d.items()}
    wc=m.get('SQ_WAVE_CYCLES',1)
    print("%-26s" % n, " ".join("%s=%.3g" % (k.replace('SQ_',''), v) for k,v in sorted(m.items())))
    print("    shares of wave cycles: VALU active %.1f%%  LDS active %.1f%%  WAIT_ANY %.1f%%  WAIT_INST_ANY %.1f%%" % (
        100*m.get('SQ_ACTIVE_INST_VALU',0)/wc, 100*m.get('SQ_ACTIVE_INST_LDS',0)/wc, 100*m.get('SQ_WAIT_ANY',0)/wc, 100*m.get('SQ_WAIT_INST_ANY',0)/wc))
PY
