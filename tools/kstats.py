"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (csv output): tools/kstats.py DIR [steps]."""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("file", f[0], " total kernel time per step %.3f ms over %d steps" % (tot / 1e6 / steps, steps))
for r in rows[:40]:
    print("%-110s calls/step %6.1f  ms/step %8.3f  avg %8.1f us" % (r["Name"][:110], float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
