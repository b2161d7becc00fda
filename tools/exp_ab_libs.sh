#!/bin/bash
# tools/exp_ab_libs.sh TAG OLDLIB workload... : bench.py --workload W with the shipped library and with OLDLIB (DSPSR_AMD_LIB),
# same box, alternating twice
T=$1; OLD=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
for rep in 1 2; do
  for w in "$@"; do
    python3 bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 --no-h2d > gpurun_out/$T/${w}_new_$rep.json 2> gpurun_out/$T/${w}_new_$rep.err || exit 1
    DSPSR_AMD_LIB=$R/$OLD python3 bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 --no-h2d > gpurun_out/$T/${w}_old_$rep.json 2> gpurun_out/$T/${w}_old_$rep.err || exit 1
  done
done
python3 - "$T" <<'PY'
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s/*_[12].json" % sys.argv[1])):
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("roofline_fused", {}).get("frac"), d["parity_gate"]["status"])
PY
