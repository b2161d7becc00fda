# second SQ pass: where the issue stalls go (LDS / VMEM queues), scalar and memory instruction cycles
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
rm -rf gpurun_out/pmc_sq2
timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/pmc_sq2.log 2>&1
f=$(find gpurun_out/pmc_sq2 -name "*counter_collection.csv" | head -1)
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
    print("%-26s" % n, " ".join("%s=%.1f%%" % (k.replace('SQ_',''), 100*v/wc) for k,v in sorted(m.items()) if k!='SQ_WAVE_CYCLES'))
PY
done
