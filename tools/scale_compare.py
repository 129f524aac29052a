"""Per-launch durations of one iteration at two shard sizes (rocprofv3 kernel traces), matched in launch order."""
import csv, sys
def last_iter(path, iters):
    rows = list(csv.DictReader(open(path))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // iters
    return [(r["Kernel_Name"].split("(")[0][-70:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-n:]]
a, b, ratio = last_iter(sys.argv[1], 3), last_iter(sys.argv[2], 3), float(sys.argv[3])
if len(a) != len(b):
    print("launch counts differ", len(a), len(b))
tot = 0.0
for (na, da), (nb, db) in zip(a, b):
    if da < 150: continue
    ex = db - da / ratio
    tot += ex
    print(f"{na:72s} big {da:9.1f} us  /{ratio:g} = {da/ratio:8.1f}  small {db:8.1f}  excess {ex:7.1f}")
print("total excess of launches > 150 us:", round(tot, 1), "us")
