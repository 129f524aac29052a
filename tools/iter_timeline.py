"""Timeline of the last training iteration in a rocprofv3 --kernel-trace CSV of bench.py: every launch of at least
MIN_US microseconds with its start (relative to the previous Adam step), duration, queue and grid, then the busy / idle
split of the iteration and the total of the shorter launches.
usage: python tools/iter_timeline.py <kernel_trace.csv> [min_us=150] [back=0]   (back: the iteration that many before the last one - bench.py's timed
region lies behind its warm-up and in front of its breakdown pass of min(steps, 10) iterations)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [r for r in rows if "adam_kernel" in r["Kernel_Name"]]
if len(adam) < 2:
    sys.exit("need two Adam steps in the trace")
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(adam) < back + 2:
    sys.exit("not that many iterations in the trace")
ta, tb = int(adam[-2 - back]["Start_Timestamp"]), int(adam[-1 - back]["Start_Timestamp"])
it = [r for r in rows if ta < int(r["Start_Timestamp"]) <= tb]
print(f"iteration: {(tb - ta) / 1e6:.3f} ms, {len(it)} launches")
for r in it:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < min_us:
        continue
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:78]
    print(f"{(int(r['Start_Timestamp']) - ta) / 1e6:8.3f} ms  +{d / 1e3:7.3f} ms  q{r['Queue_Id']}  {name:78s} grid {r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in it)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
short = [r for r in it if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 < min_us]
print(f"device busy {busy / 1e6:.3f} ms, idle {(tb - ta - busy) / 1e6:.3f} ms; {len(short)} launches shorter than {min_us:.0f} us: "
      f"{sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in short) / 1e6:.3f} ms in all (partly on side streams)")
