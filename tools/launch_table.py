"""Per-launch table of one training iteration from a rocprofv3 --kernel-trace CSV (tuning aid).
usage: python tools/launch_table.py <kernel_trace.csv> [iterations_in_trace]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // iters
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < float(sys.argv[3] if len(sys.argv) > 3 else 100): continue
    name = r["Kernel_Name"]
    short = name.split("(")[0][:90]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:9.3f} ms  {d:10.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?')):>4}  {short}")
