#!/bin/bash
# tools/exp_fold_ab.sh TAG OLDLIB : the fold-only benchmark (Benchmark/fold.csh shape) with the shipped library and with an
# older build (DSPSR_AMD_LIB), same box, alternating.
T=${1:-foldab}; OLD=$2
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
for i in 1 2; do
  python3 bench.py --workload fold --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/$T/new_$i.json 2> gpurun_out/$T/new_$i.err || exit 1
  DSPSR_AMD_LIB=$R/$OLD python3 bench.py --workload fold --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/$T/old_$i.json 2> gpurun_out/$T/old_$i.err || exit 1
done
python3 - "$T" <<'PY'
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s/*.json" % sys.argv[1])):
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity_gate"]["status"])
PY
