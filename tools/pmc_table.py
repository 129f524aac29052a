"""Per-dispatch table of a rocprofv3 --pmc ... --kernel-trace run: duration, every collected counter, and the derived
shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and MfmaUtil (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)).  usage: pmc_table.py DIR [name filter]"""
import csv, glob, os, sys
from collections import OrderedDict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
rows = OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        if flt and flt not in r["Kernel_Name"]:
            continue
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        e = rows.setdefault(key, {"t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for e in rows.values() for k in e if k != "t"})
print("%-4s %-58s %9s " % ("id", "kernel", "ms") + " ".join("%22s" % n for n in names) + "   derived")
for (did, kn), e in sorted(rows.items()):
    short = kn.replace("void dgp::", "").split("(")[0][:58]
    der = []
    if "GRBM_GUI_ACTIVE" in e:
        der.append("clock %.3f GHz" % (e["GRBM_GUI_ACTIVE"] / 8 / (e["t"] * 1e6)))      # GUI_ACTIVE sums the 8 XCDs
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        # rocprofv3's MfmaUtil: busy cycles summed over the 1024 SIMDs / (active cycles per XCD x 1024)
        der.append("MfmaUtil %.3f" % (e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    print("%-4d %-58s %9.3f " % (did, short, e["t"]) + " ".join("%22.0f" % e.get(n, float("nan")) for n in names) + "   " + "; ".join(der))
