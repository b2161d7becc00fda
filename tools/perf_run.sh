#!/bin/bash
# tools/perf_run.sh TAG [bench args] : on the GPU box -- rocprofv3 kernel stats of one bench.py run (no CPU baseline),
# per-kernel averages and the bench line under gpurun_out/TAG/
T=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof -- python bench.py --no-cpu-baseline "$@" > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err || { echo "bench failed"; tail -5 gpurun_out/$T/bench.err; exit 1; }
python tools/kstats.py gpurun_out/$T/prof | tee gpurun_out/$T/kstats.txt
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$T/bench.json") if l.startswith("{")][0])
print("value %.0f  ms/step %.4f  frac %.4f  fused_frac %s  gate %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("roofline_fused",{}).get("frac"), d.get("parity_gate",{}).get("status")))
PY
