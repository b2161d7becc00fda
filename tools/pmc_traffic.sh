#!/bin/bash
# tools/pmc_traffic.sh TAG : HBM traffic per kernel of the default bench.py launch shape from the PMC counters, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes (--pmc with --kernel-trace only), reads = 2 x
# FETCH_SIZE (gfx950 counts 64 B per 128-B request), KB = 1024 B.  Writes gpurun_out/TAG/traffic.json (copy to profiles/).
T=$1
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/$T/pmc_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/$T/pmc_$c -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-companions > gpurun_out/$T/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/$T/pmc_$c.log; exit 1; }
done
python - "$T" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
mean = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/%s/pmc_%s/**/*counter_collection.csv" % (tag, c), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "dspsr_amd" not in n:
            continue
        acc[n.split("(")[0].replace("void dspsr_amd::", "").replace("dspsr_amd::", "")].append(float(r["Counter_Value"]))
    mean[c] = {k: sum(v) / len(v) for k, v in acc.items()}
line = json.loads([l for l in open("gpurun_out/%s/pmc_FETCH_SIZE.log" % tag) if l.startswith("{")][0])
mp, nkeep = line["config"]["max_parts"], line["config"]["nkeep"]
sub = max(1, min(mp, (512 << 20) // (line["config"]["n_fft"] * 2 * 8)))       # parts per pass-2 / inverse sub-group
ker = {}
for k in mean["FETCH_SIZE"]:
    # passes 0 and 1 run once per launch group, pass 2 / inverse once per sub-group, the stand-alone fold once per BLOCK
    parts = mp if ("raw_transpose" in k or "fwd_cols" in k) else line["config"]["parts_per_block"] if "fold_chunked" in k else sub
    ker[k] = {"fetch_KB": round(mean["FETCH_SIZE"][k], 1), "write_KB": round(mean["WRITE_SIZE"].get(k, 0.0), 1), "parts": parts,
              "MB_per_part": round((2 * mean["FETCH_SIZE"][k] + mean["WRITE_SIZE"].get(k, 0.0)) * 1024 / parts / 1e6, 2)}
def per_part(pred):
    return sum(v["MB_per_part"] for k, v in ker.items() if pred(k))
common = lambda k: "fold_chunked" not in k and "inv_chan" not in k
unf = per_part(common) + per_part(lambda k: "inv_chan" in k and "false" in k)
fus = per_part(common) + per_part(lambda k: "inv_chan" in k and "true" in k)
out = {"workload": line["config"]["workload"], "parts_per_launch_group": mp,
       "source": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; reads = 2 x "
                 "FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section); KB = 1024 B; pass 2 and the inverse pass are "
                 "dispatched per %d parts, passes 0 and 1 per %d" % (sub, mp),
       "kernels": ker, "hbm_MB_per_part": round(unf, 1), "hbm_MB_per_part_fused": round(fus, 1),
       "hbm_bytes_per_launch_group": int(unf * 1e6 * mp), "hbm_bytes_per_launch_group_fused": int(fus * 1e6 * mp),
       "algorithmic_MB_per_part": round(line["roofline"]["algorithmic_bytes_per_part"] / 1e6, 2)}
json.dump(out, open("gpurun_out/%s/traffic.json" % tag, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_MB_per_part", "hbm_MB_per_part_fused", "algorithmic_MB_per_part")}))
for k, v in ker.items():
    print("%-30s %s" % (k, v))
PY
