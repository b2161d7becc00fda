#!/bin/bash
# tools/exp_bench4.sh TAG [workloads...] : bench.py --workload W (no CPU baseline, no PCIe leg), twice each, values printed
T=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
for i in 1 2; do
for w in "$@"; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 --no-h2d > gpurun_out/$T/${w}_$i.json 2> gpurun_out/$T/${w}_$i.err || { echo "$w failed"; tail -3 gpurun_out/$T/${w}_$i.err; exit 1; }
done
done
python3 - "$T" <<'PY'
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s/*_[12].json" % sys.argv[1])):
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("roofline_fused", {}).get("frac"), d["parity_gate"]["status"])
PY
