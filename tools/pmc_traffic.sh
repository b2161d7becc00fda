#!/bin/bash
# tools/pmc_traffic.sh TAG WORKLOAD [K] : HBM-side traffic of one bench.py workload from the PMC counters, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes (--pmc with --kernel-trace only), reads = 2 x
# FETCH_SIZE (gfx950 counts 64 B per 128-B request), KB = 1024 B.  The profiled program is tools/pmc_workload.py: K identical
# blocks in the bench's launch shape, fused and (fold workloads) unfused.  Writes gpurun_out/TAG/WORKLOAD_traffic.json
# (copy to profiles/rNN_WORKLOAD_traffic.json; bench.py reads the newest one per workload).
T=$1
W=${2:-target}
K=${3:-4}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
MODES="fused unfused"
if [ "$W" = cfg5 ] || [ "$W" = cfg5c ] || [ "$W" = fold ] || [ "$W" = after ] || [ "$W" = after8k ] || [ "$W" = after1k ] || [ "$W" = after8c ] || [ "$W" = plain ]; then MODES="unfused"; fi
for m in $MODES; do
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/$T/pmc_${W}_${m}_$c
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/pmc_workload.py $W $m $K > $d.log 2>&1 || { echo "pass $W $m $c failed"; tail -5 $d.log; exit 1; }
done
done
python3 - "$T" "$W" "$K" "$MODES" <<'PY'
import collections, csv, glob, json, subprocess, sys
tag, wl, K, modes = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4].split()
# torch kernels that only make the synthetic block (its chunks are assembled with device-to-device copies; the pipeline itself
# issues none for a resident block: fold plans travel by hipMemcpyAsync from pinned memory, which is not a kernel)
skip = ("at::", "at_cuda", "elementwise", "distribution", "philox", "Cijk", "copyBuffer")
out = {"workload": wl, "blocks": K,
       "source": "tools/pmc_traffic.sh (tools/pmc_workload.py: K identical blocks in bench.py's launch shape): rocprofv3 --pmc FETCH_SIZE / "
                 "--pmc WRITE_SIZE in separate passes with --kernel-trace only; reads = 2 x FETCH_SIZE (gfx950 correction, "
                 "MI355X_MICROARCH.md HBM section); KB = 1024 B; every kernel of the run except torch's generators and copies, memsets "
                 "included; the unfused group is the filterbank launch group bench.py's `roofline` prices (FFT + chirp + detect, "
                 "detected output written): the stand-alone Fold kernels that follow it are listed under `kernels` and summed in "
                 "`fold_MB_per_part`, not in the group"}
for m in modes:
    tot, foldb = {}, 0.0
    ker = collections.defaultdict(lambda: {"calls_per_block": 0.0, "fetch_KB_per_block": 0.0, "write_KB_per_block": 0.0})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("gpurun_out/%s/pmc_%s_%s_%s/**/*counter_collection.csv" % (tag, wl, m, c), recursive=True)[0]
        s = 0.0
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if any(x in n for x in skip):
                continue
            k = n.split("(")[0].replace("void dspsr_amd::", "").replace("dspsr_amd::", "").strip()
            v = float(r["Counter_Value"])
            if m == "unfused" and "k_fold_" in k and wl not in ("fold",):
                foldb += v * (2 if c == "FETCH_SIZE" else 1) * 1024 / K          # stand-alone Fold behind the priced group
            else:
                s += v
            e = ker[k]
            if c == "FETCH_SIZE":
                e["calls_per_block"] += 1.0 / K
                e["fetch_KB_per_block"] += v / K
            else:
                e["write_KB_per_block"] += v / K
        tot[c] = s / K
    line = json.loads([l for l in open("gpurun_out/%s/pmc_%s_%s_FETCH_SIZE.log" % (tag, wl, m)) if l.startswith("{")][-1])
    for e in ker.values():
        e["MB_per_block"] = round((2 * e["fetch_KB_per_block"] + e["write_KB_per_block"]) * 1024 / 1e6, 2)
        for q in ("calls_per_block", "fetch_KB_per_block", "write_KB_per_block"):
            e[q] = round(e[q], 2)
    hbm = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
    sfx = "_fused" if m == "fused" else ""
    ppb, mp = line["parts_per_block"], line["parts_per_launch_group"]
    nch = round(line["algorithmic_bytes_per_block"] / line["algorithmic_bytes_per_part"] / ppb) if "algorithmic_bytes_per_part" in line else 1
    out["parts_per_block"], out["parts_per_launch_group"] = ppb, mp
    out["hbm_bytes_per_block" + sfx] = int(hbm)
    out["hbm_MB_per_part" + sfx] = round(hbm / (ppb * nch) / 1e6, 2)
    out["hbm_bytes_per_launch_group" + sfx] = int(hbm / (ppb * nch) * mp)
    out["algorithmic_bytes_per_block" + sfx] = line["algorithmic_bytes_per_block"]
    out["algorithmic_MB_per_part" + sfx] = round(line["algorithmic_bytes_per_block"] / (ppb * nch) / 1e6, 2)
    out["traffic_ratio" + sfx] = round(hbm / line["algorithmic_bytes_per_block"], 3)
    if foldb:
        out["fold_MB_per_part"] = round(foldb / (ppb * nch) / 1e6, 2)
    out["kernels" + sfx] = dict(ker)
    if "roofline_kernel" in line:                       # cfg5 / fold: the dominant kernel on its own, per launch
        k = [n for n in ker if line["roofline_kernel"] in n]
        if k:
            e = ker[k[0]]
            out["hbm_bytes_per_launch"] = int((2 * e["fetch_KB_per_block"] + e["write_KB_per_block"]) * 1024 / max(e["calls_per_block"], 1e-9))
            out["roofline_kernel"] = k[0]
# the library the counters were collected on (the GPU box has no .git: the id is baked into the library at make time)
out["build_id"] = line.get("build_id")
json.dump(out, open("gpurun_out/%s/%s_traffic.json" % (tag, wl), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("kernels") and k != "source"}))
for m in modes:
    for k, v in out["kernels" + ("_fused" if m == "fused" else "")].items():
        print("%-8s %-34s %s" % (m, k, v))
PY
