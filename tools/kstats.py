"""Average duration per kernel name (and grid) of the LAST `n` iterations in a rocprofv3 --kernel-trace CSV of bench.py.
usage: python tools/kstats.py <kernel_trace.csv> [substring ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
rows = rows[adam[1]:adam[-1]] if len(adam) > 2 else rows
nit = max(1, len(adam) - 2)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("dgp::", "").split("(")[0][:60] + " g" + r["Grid_Size_X"] + "," + r["Grid_Size_Y"]
    a = acc[name]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    print(f"{us / nit / 1e3:8.3f} ms/iter  {n / nit:5.1f} launches/iter  {us / n:9.1f} us avg  {name}")
