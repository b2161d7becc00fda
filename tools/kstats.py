"""Print per-kernel rocprofv3 stats (dspsr_amd kernels only) from a *kernel_stats.csv, names trimmed."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "dspsr_amd" not in n:
        continue
    n = n.split("(")[0].replace("void dspsr_amd::", "")
    print("%-28s calls=%-4s avg_us=%8.1f  pct=%s" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
