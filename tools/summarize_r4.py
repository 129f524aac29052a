#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_r4.sh into the small summaries kept under profiles/.

usage: summarize_r4.py <collect dir> <out dir>
Writes r4_kernel_stats.csv (the --stats table), r4_pmc_fetch_size.csv / r4_pmc_write_size.csv (per-kernel-name sums
of the counters), r4_pmc_mfma.json (matrix-pipe counters of the GEMM engine: executed fp64 MFMA flops, MFMA busy
fraction, effective clock) and r4_pmc_traffic.json (HBM bytes per launch of the GEMM engine, corrected as
MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE is in KB and gfx950 reports half of wide coalesced reads).
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


shutil.copy(one("stats/**/*_kernel_stats.csv"), os.path.join(out, "r4_kernel_stats.csv"))


def counter_sums(path, counter):
    sums, calls = defaultdict(float), defaultdict(int)
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            sums[row["Kernel_Name"]] += float(row["Counter_Value"])
            key = (row["Kernel_Name"], row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[row["Kernel_Name"]] += 1
    return sums, calls


fetch, fcalls = counter_sums(one("pmc_fetch/**/*_counter_collection.csv"), "FETCH_SIZE")
write, wcalls = counter_sums(one("pmc_write/**/*_counter_collection.csv"), "WRITE_SIZE")
for name, table, calls in (("fetch", fetch, fcalls), ("write", write, wcalls)):
    with open(os.path.join(out, f"r4_pmc_{name}_size.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", f"{name.upper()}_SIZE_KB_sum"])
        for k in sorted(table, key=table.get, reverse=True):
            w.writerow([k, calls[k], f"{table[k]:.3f}"])

gemm = [k for k in fetch if "gemm_f64_kernel" in k or "gemm_wide_kernel" in k or "gemm_tall_kernel" in k or "gemm_tallu_kernel" in k or "gemm_gram_kernel" in k or "g_panel_kernel" in k]
launches = sum(fcalls[k] for k in gemm)
# iterations in the traced run = Adam steps (bench.py --steps 2 --warmup 1 runs 1 + 2 iterations and a breakdown pass of 2 more)
steps = sum(n for k, n in fcalls.items() if "adam_kernel" in k) or 3
# bench.py's roofline block counts the point contractions only (launches_per_step); the M^3 launches of the small-matrix
# chain move a few MB each, so the per-launch traffic is quoted over the same launches as `achieved`
try:
    point_launches = json.load(open(os.path.join(src, "stats_bench.json")))["roofline"]["launches_per_step"]
except Exception:
    point_launches = None
total = sum(fetch[k] for k in gemm) * 1024 * 2 + sum(write.get(k, 0.0) for k in gemm) * 1024
all_fetch = sum(fetch.values()) * 1024 * 2 + sum(write.values()) * 1024
import subprocess
try:
    commit = open(os.path.join(src, "commit.txt")).read().strip()
except Exception:
    commit = "unknown"
json.dump({
    "commit": commit,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 (+ its breakdown pass: 5 iterations in all), config 2, 1 GPU",
    "correction": "FETCH_SIZE (KB) x 1024 x 2 (gfx950 halves wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE (KB) x 1024",
    "kernel": "dgp::gemm_wide_kernel + dgp::gemm_tall_kernel + dgp::gemm_tallu_kernel + dgp::gemm_gram_kernel + dgp::g_panel_kernel + dgp::gemm_f64_kernel (all instantiations)",
    "launches": launches,
    "hbm_bytes_total": total,
    "hbm_bytes_per_launch": (total / steps / point_launches) if point_launches else total / max(launches, 1),
    "point_contraction_launches_per_step": point_launches,
    "hbm_bytes_per_step": total / steps,
    "all_kernels_hbm_bytes_per_step": all_fetch / steps,
}, open(os.path.join(out, "r4_pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, "r4_pmc_traffic.json")).read())


# ---- matrix-pipe counters of the GEMM engine (third PMC pass) ----
mfma_csv = glob.glob(os.path.join(src, "pmc_mfma/**/*_counter_collection.csv"), recursive=True)
if mfma_csv:
    per = defaultdict(lambda: defaultdict(float))          # dispatch -> counter -> value
    names = {}
    with open(mfma_csv[0], newline="") as f:
        for row in csv.DictReader(f):
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
    dur = {}
    trace = glob.glob(os.path.join(src, "pmc_mfma/**/*_kernel_trace.csv"), recursive=True)
    if trace:
        with open(trace[0], newline="") as f:
            for row in csv.DictReader(f):
                dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    SIMDS, XCDS = 1024, 8
    busy = mops = gui = ns = 0.0
    big = []
    for d, c in per.items():
        if "gemm_f64_kernel" not in names[d] and "gemm_wide_kernel" not in names[d] and "gemm_tall_kernel" not in names[d] and "gemm_tallu_kernel" not in names[d] and "gemm_gram_kernel" not in names[d] and "g_panel_kernel" not in names[d]:
            continue
        busy += c["SQ_VALU_MFMA_BUSY_CYCLES"]; mops += c["SQ_INSTS_VALU_MFMA_MOPS_F64"]; gui += c["GRBM_GUI_ACTIVE"]
        ns += dur.get(d, 0)
        if dur.get(d, 0) > 4e6:
            big.append({"kernel": names[d].split("(")[0][-60:], "ms": dur[d] / 1e6,
                        "executed_tflops": c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / dur[d] / 1e3,
                        "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / XCDS * SIMDS),
                        "clock_ghz": c["GRBM_GUI_ACTIVE"] / XCDS / dur[d]})
    json.dump({
        "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace, bench.py --steps 2 --warmup 1 (+ its breakdown pass: 5 iterations in all), config 2, 1 GPU",
        "kernel": "dgp::gemm_wide_kernel + dgp::gemm_tall_kernel + dgp::gemm_tallu_kernel + dgp::gemm_gram_kernel + dgp::g_panel_kernel + dgp::gemm_f64_kernel (all instantiations, all launches of the 3 iterations)",
        "formulas": "executed flops = MOPS_F64 x 512; MfmaUtil = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs) (rocprofv3's own MfmaUtil expression); clock = GRBM_GUI_ACTIVE/8/duration",
        "executed_mfma_flops_per_step": mops * 512 / steps,
        "executed_tflops_over_gemm_time": mops * 512 / ns / 1e3 if ns else None,
        "mfma_busy_frac": busy / (gui / XCDS * SIMDS) if gui else None,
        "effective_clock_ghz": gui / XCDS / ns if ns else None,
        "gemm_ms_per_step_profiled": ns / 1e6 / steps,
        "launches_over_5ms_last_first": big[-12:],
    }, open(os.path.join(out, "r4_pmc_mfma.json"), "w"), indent=1)
    print(open(os.path.join(out, "r4_pmc_mfma.json")).read())


# ---- LDS counters of the GEMM engine (fourth PMC pass) ----
lds_csv = glob.glob(os.path.join(src, "pmc_lds/**/*_counter_collection.csv"), recursive=True)
if lds_csv:
    tot = defaultdict(float)
    with open(lds_csv[0], newline="") as f:
        for row in csv.DictReader(f):
            if "gemm_f64_kernel" in row["Kernel_Name"] or "gemm_wide_kernel" in row["Kernel_Name"] or "gemm_tall_kernel" in row["Kernel_Name"] or "gemm_tallu_kernel" in row["Kernel_Name"] or "gemm_gram_kernel" in row["Kernel_Name"] or "g_panel_kernel" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    act, conf, gui = tot["SQ_LDS_IDX_ACTIVE"], tot["SQ_LDS_BANK_CONFLICT"], tot["GRBM_GUI_ACTIVE"]
    json.dump({
        "source": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace, bench.py --steps 2 --warmup 1 (+ its breakdown pass: 5 iterations in all), config 2, 1 GPU",
        "kernel": "dgp::gemm_wide_kernel + dgp::gemm_tall_kernel + dgp::gemm_tallu_kernel + dgp::gemm_gram_kernel + dgp::g_panel_kernel + dgp::gemm_f64_kernel (all instantiations)",
        "formulas": "bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent re-issuing conflicting lanes over the cycles it was busy); LDS busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE/8 XCDs x 256 CUs) (rocprofv3's LdsUtil expression)",
        "lds_bank_conflict_share": conf / act if act else None,
        "lds_busy_frac": act / (gui / 8 * 256) if gui else None,
        "raw": dict(tot),
    }, open(os.path.join(out, "r4_pmc_lds.json"), "w"), indent=1)
    print(open(os.path.join(out, "r4_pmc_lds.json")).read())
