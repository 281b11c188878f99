"""Folds the output of tools/profile_r03.sh (gpurun_out/r03) into profiles/: one r03_<case>_kernel_stats.csv per case
(rocprofv3's own per-kernel summary), profiles/r03_pmc_traffic.csv (HBM bytes per launch and kernel), profiles/r03_summary.csv
(time, algorithmic bytes, measured traffic and roofline fraction per case), profiles/r03_qp_long_contract_pmc.csv (the
matrix-core counters of the contraction kernel) and the matching entries of profiles/traffic.json, which bench.py reports
as roofline.traffic.  A case names the kernels one step of the operation launches; its time and traffic are their sums.
WRITE_SIZE and FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, hence the
factor 2 on the read side."""
import csv, glob, json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r03")
PROF = os.path.join(ROOT, "profiles")
# case -> (traffic.json key, regex of the kernels of one step)
CASES = {
    "qp_two_mass": ("qp:two_mass:1048576:20:20", r"qp_lane_kernel"),
    "qp_msd": ("qp:msd:65536:20:20", r"qp_lane_kernel"),
    "qp_two_mass_f32": ("qp:two_mass:1048576:32:32", r"qp_lane_kernel"),
    "qp_two_mass_held": ("qp:two_mass:1048576:30:10", r"qp_lane_kernel"),
    "qp_quadrotor": ("qp:quadrotor:16384:50:50", r"qp_wg_kernel|w3_band_kernel"),
    "qp_long": ("qp:long:1024:400:400", r"qp_wg_kernel|w3_band_kernel"),
    "qp_quadrotor_held": ("qp:quadrotor:16384:50:25", r"qp_wg_kernel|w3_band_kernel"),
    "qp_long_held": ("qp:long:1024:400:200", r"qp_wg_kernel|w3_band_kernel"),
    "qp_long_contract": (None, r"hessian_mfma_kernel"),
    "rollout_msd": ("rollout:msd:65536:20:20", r"rollout_lane_kernel"),
    "rollout_two_mass": ("rollout:two_mass:1048576:20:20", r"rollout_lane_kernel"),
    "rollout_quadrotor": ("rollout:quadrotor:16384:50:50", r"rollout_group_kernel|rollout_quad_kernel"),
    "solve_two_mass": ("solve:two_mass:1048576:20:20", r"solve_rows2?_kernel"),
    "gain_two_mass": ("gain:two_mass:1048576:20:20", r"solve_rows2?_kernel"),
    "gain_rows_two_mass": ("gain_rows:two_mass:1048576:20:20", r"solve_rows2?_kernel|rows_lane_kernel|unit_columns_kernel"),
    "loop_two_mass": ("loop:two_mass:1048576:20:20", r"closed_loop_lane_kernel"),
    "loop_quadrotor": ("loop:quadrotor:16384:50:50", r"closed_loop_affine_kernel|closed_loop_rows_kernel|closed_loop_group_kernel"),
    "gain_quadrotor": ("gain:quadrotor:16384:50:50", r"lu_\w+"),
    "gain_rows_quadrotor": ("gain_rows:quadrotor:16384:50:50", r"lu_\w+|rows_wg_kernel|unit_columns_kernel"),
    "gain_long": ("gain:long:1024:400:400", r"lu_\w+"),
    "gain_rows_long": ("gain_rows:long:1024:400:400", r"lu_\w+|rows_wg_kernel|unit_columns_kernel"),
}


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return m.group(0) if m else name[:60]


table = json.load(open(os.path.join(PROF, "traffic.json")))
pmc_rows, summary = [], []
for case, (key, pat) in CASES.items():
    ks = glob.glob(os.path.join(SRC, f"kt_{case}", "**", "*kernel_stats.csv"), recursive=True)
    jf = os.path.join(SRC, f"kt_{case}.json")
    if not ks or not os.path.exists(jf):
        continue
    lines = [l for l in open(jf) if l.startswith("{")]
    if not lines:
        continue
    bench = json.loads(lines[-1])
    shutil.copy(ks[0], os.path.join(PROF, f"r03_{case}_kernel_stats.csv"))
    launches = bench["steps"] + bench["warmup"]
    kern = {}
    for row in csv.DictReader(open(ks[0])):
        if re.search(pat, row["Name"]):
            kern[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
    # time of one step: each matching kernel's average duration times its launches per step (the contraction case also
    # times the kernel on its own, so its calls are not a multiple of the steps: one launch per step)
    step_ms = 0.0
    for name, (calls, avg) in kern.items():
        per_step = 1 if case == "qp_long_contract" else max(1, round(calls / launches))
        step_ms += avg * per_step / 1e6
    traffic = {}
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        files = glob.glob(os.path.join(SRC, f"pmc_{case}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        pj = os.path.join(SRC, f"pmc_{case}_{ctr}.json")
        if not files or not os.path.exists(pj):
            continue
        pl = [l for l in open(pj) if l.startswith("{")]
        if not pl:
            continue
        pb = json.loads(pl[-1])
        nl = pb["steps"] + pb["warmup"]
        per = {}
        for row in csv.DictReader(open(files[0])):
            if re.search(pat, row["Kernel_Name"]) and row["Counter_Name"] == ctr:
                per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
        tot = 0.0
        for name, vals in per.items():
            per_step = 1 if case == "qp_long_contract" else max(1, round(len(vals) / nl))
            tot += sum(vals) / len(vals) * per_step
            pmc_rows.append((case, short(name), ctr, len(vals), sum(vals) / len(vals)))
        traffic[ctr] = tot
    alg = bench["roofline"].get("bytes_per_unit", 0) * bench["config"]["batch_per_gpu"]
    w = traffic.get("WRITE_SIZE", float("nan")) * 1024
    r = 2 * traffic.get("FETCH_SIZE", float("nan")) * 1024
    if key and len(traffic) == 2:
        table[key] = {"hbm_bytes_per_launch": w + r, "write_bytes": w, "read_bytes": r, "algorithmic_bytes": alg,
                      "source": "profiles/r03_pmc_traffic.csv",
                      "note": "kernels of one step (%s): WRITE_SIZE x 1024 + 2 x FETCH_SIZE x 1024, summed over the step's kernels, mean over steps" % pat}
    names = " + ".join(sorted(short(n) for n in kern))
    summary.append((case, names, step_ms, bench["ms_per_step"], alg, w + r))

if os.path.isdir(os.path.join(SRC, "pmc_qp_long_contract_MFMA")):
    files = glob.glob(os.path.join(SRC, "pmc_qp_long_contract_MFMA", "**", "*counter_collection.csv"), recursive=True)
    if files:
        acc = {}
        for row in csv.DictReader(open(files[0])):
            if "hessian_mfma_kernel" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        with open(os.path.join(PROF, "r03_qp_long_contract_pmc.csv"), "w") as fh:
            fh.write("kernel,counter,dispatches,mean_value_per_dispatch\n")
            for k, v in sorted(acc.items()):
                fh.write("hessian_mfma_kernel,%s,%d,%.1f\n" % (k, len(v), sum(v) / len(v)))
                print("hessian_mfma_kernel %-32s %3d dispatches, mean %.1f" % (k, len(v), sum(v) / len(v)))

json.dump(table, open(os.path.join(PROF, "traffic.json"), "w"), indent=1)
with open(os.path.join(PROF, "r03_pmc_traffic.csv"), "w") as fh:
    fh.write("bench,kernel,counter,dispatches,mean_value_KiB\n")
    for r_ in pmc_rows:
        fh.write("%s,\"%s\",%s,%d,%.4f\n" % r_)
with open(os.path.join(PROF, "r03_summary.csv"), "w") as fh:
    fh.write("case,kernels,kernel_ms_per_step,bench_ms_per_step,algorithmic_GB,pmc_traffic_GB,traffic_over_algorithmic,algorithmic_TB_per_s,frac_of_8TBps\n")
    for case, names, ms, bms, alg, tr in summary:
        fh.write(f"{case},\"{names}\",{ms:.4f},{bms:.4f},{alg/1e9:.4f},{tr/1e9:.4f},{(tr/alg if alg else float('nan')):.4f},{(alg/ms/1e9 if ms else 0):.3f},{(alg/ms/1e9/8 if ms else 0):.3f}\n")
        print(f"{case:22s} {ms:9.4f} ms (bench {bms:9.4f})  alg {alg/1e9:8.3f} GB  pmc {tr/1e9:9.3f} GB ({(tr/alg if alg else float('nan')):.3f}x)  {(alg/ms/1e9/8 if ms else 0):.3f} of 8 TB/s   {names[:70]}")
