#!/bin/bash
# tools/exp_cfg4_ab.sh TAG : the cfg4 sub-band shard through the two-pass path and through the three-pass kernels, same box,
# alternating, then a kernel trace of the two-pass run.  Output under gpurun_out/TAG/.
T=${1:-cfg4ab}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/$T
for i in 1 2; do
  python3 bench.py --workload cfg4 --no-cpu-baseline --steps 40 --warmup 5 > $O/cfg4_two_$i.json 2> $O/cfg4_two_$i.err || exit 1
  python3 bench.py --workload cfg4 --no-cpu-baseline --steps 40 --warmup 5 --no-two-pass > $O/cfg4_three_$i.json 2> $O/cfg4_three_$i.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python3 bench.py --workload cfg4 --no-cpu-baseline --steps 20 --warmup 3 > $O/prof_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4_three -- python3 bench.py --workload cfg4 --no-cpu-baseline --steps 20 --warmup 3 --no-two-pass > $O/prof_cfg4_three.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
for f in sorted(glob.glob(o + "/cfg4_*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_fused"]["frac"], d["parity_gate"]["status"])
for f in sorted(glob.glob(o + "/prof_cfg4*/**/*kernel_stats.csv", recursive=True)):
    print(f)
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print("%-60s calls=%-5s avg_us=%9.1f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
