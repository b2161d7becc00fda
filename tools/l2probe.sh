# runs the L2 exchange probe: timing for all modes, then HBM-side traffic (PMC) per mode and scratch size
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=build/l2probe
for ck in 64 32; do
for m in 3 0 1 2 4 5; do timeout -k 10 60 $P $m 200 2 $ck || exit 1; done
for m in 0 1 4 5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/l2p_$m$c
    timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/l2p_$m$c -- $P $m 200 1 $ck > /dev/null 2>&1
    f=$(find gpurun_out/l2p_$m$c -name "*counter_collection.csv" | head -1)
    python3 -c "
import csv,sys
v=[float(r['Counter_Value']) for r in csv.DictReader(open('$f')) if 'k_probe' in r['Kernel_Name']]
print('mode $m chunk $ck KB $c: %.1f KB per launch = %.1f KB/iter' % (sum(v)/len(v), sum(v)/len(v)/200))"
  done
done
done
