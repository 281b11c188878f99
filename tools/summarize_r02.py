"""Folds the output of tools/profile_r02.sh (gpurun_out/r02) into profiles/: one r02_<case>_kernel_stats.csv per case
(rocprofv3's own per-kernel summary), profiles/r02_pmc_traffic.csv (HBM bytes per launch of each case's dominant kernel)
and the matching entries of profiles/traffic.json, which bench.py reports as roofline.traffic.
WRITE_SIZE and FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, hence the
factor 2 on the read side."""
import csv, glob, json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02")
PROF = os.path.join(ROOT, "profiles")
# case -> (traffic.json key, dominant kernel)
CASES = {
    "qp_two_mass": ("qp:two_mass:1048576:20:20", "qp_lane_kernel"),
    "qp_msd": ("qp:msd:65536:20:20", "qp_lane_kernel"),
    "qp_quadrotor": ("qp:quadrotor:16384:50:50", "qp_wg_kernel"),
    "qp_long": ("qp:long:1024:400:400", "qp_wg_kernel"),
    "qp_quadrotor_held": ("qp:quadrotor:16384:50:25", "qp_wg_kernel"),
    "qp_long_held": ("qp:long:1024:400:200", "qp_wg_kernel"),
    "rollout_msd": ("rollout:msd:65536:20:20", "rollout_lane_kernel"),
    "rollout_two_mass": ("rollout:two_mass:1048576:20:20", "rollout_lane_kernel"),
    "rollout_quadrotor": ("rollout:quadrotor:16384:50:50", "rollout_group_kernel"),
}
table = json.load(open(os.path.join(PROF, "traffic.json")))
rows, report = [], []
for case, (key, kern) in CASES.items():
    ks = glob.glob(os.path.join(SRC, f"kt_{case}", "**", "*kernel_stats.csv"), recursive=True)
    avg_ns = None
    if ks:
        shutil.copy(ks[0], os.path.join(PROF, f"r02_{case}_kernel_stats.csv"))
        for row in csv.DictReader(open(ks[0])):
            if kern in row["Name"]:
                avg_ns = float(row["AverageNs"])
                calls = int(row["Calls"])
    vals = {}
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        files = glob.glob(os.path.join(SRC, f"pmc_{case}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        acc = []
        for f in files[:1]:
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    acc.append(float(row["Counter_Value"]))
        if acc:
            vals[ctr] = sum(acc) / len(acc)
            rows.append((case, kern, ctr, len(acc), vals[ctr]))
    if len(vals) == 2:
        bench = json.loads([l for l in open(os.path.join(SRC, f"pmc_{case}_WRITE_SIZE.json")) if l.startswith("{")][-1])
        w, r = vals["WRITE_SIZE"] * 1024, 2 * vals["FETCH_SIZE"] * 1024
        alg = bench["roofline"]["bytes_per_unit"] * bench["config"]["batch_per_gpu"]
        table[key] = {"hbm_bytes_per_launch": w + r, "write_bytes": w, "read_bytes": r, "algorithmic_bytes": alg,
                      "source": "profiles/r02_pmc_traffic.csv",
                      "note": f"{kern}: WRITE_SIZE x 1024 + 2 x FETCH_SIZE x 1024, mean over its launches"}
        report.append((case, kern, avg_ns, alg, w + r))
# The blocked solver runs three kernels per launch of the gain: every one of them is listed, and the gain's traffic is their sum.
SOLVER = {"gain_long": "gain:long:1024:400:400", "gain_quadrotor": "gain:quadrotor:16384:50:50"}
solver_rows = []
for case, key in SOLVER.items():
    ks = glob.glob(os.path.join(SRC, f"kt_{case}", "**", "*kernel_stats.csv"), recursive=True)
    if not ks:
        continue
    shutil.copy(ks[0], os.path.join(PROF, f"r02_{case}_kernel_stats.csv"))
    times = {row["Name"]: (int(row["Calls"]), float(row["AverageNs"])) for row in csv.DictReader(open(ks[0])) if "lu_" in row["Name"]}
    pm = {}
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        files = glob.glob(os.path.join(SRC, f"pmc_{case}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        for f in files[:1]:
            for row in csv.DictReader(open(f)):
                if "lu_" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    pm.setdefault(row["Kernel_Name"], {}).setdefault(ctr, []).append(float(row["Counter_Value"]))
    tot_w = tot_r = tot_ms = 0.0
    for name, (calls, ns) in sorted(times.items(), key=lambda kv: -kv[1][1]):
        mt = re.search(r"lu_\w+(<[^>]*>)?", name)
        short = mt.group(0) if mt else name[:60]
        v_ = pm.get(name, {})
        w = sum(v_.get("WRITE_SIZE", [0])) / max(1, len(v_.get("WRITE_SIZE", [0]))) * 1024
        r = 2 * sum(v_.get("FETCH_SIZE", [0])) / max(1, len(v_.get("FETCH_SIZE", [0]))) * 1024
        solver_rows.append((case, short, calls, ns / 1e6, w / 1e9, r / 1e9))
        tot_w, tot_r, tot_ms = tot_w + w, tot_r + r, tot_ms + ns / 1e6
    jf = os.path.join(SRC, f"pmc_{case}_WRITE_SIZE.json")
    if os.path.exists(jf) and tot_w + tot_r > 0:
        bench = json.loads([l for l in open(jf) if l.startswith("{")][-1])
        alg = bench["roofline"]["bytes_per_unit"] * bench["config"]["batch_per_gpu"]
        table[key] = {"hbm_bytes_per_launch": tot_w + tot_r, "write_bytes": tot_w, "read_bytes": tot_r, "algorithmic_bytes": alg,
                      "source": "profiles/r02_solver_summary.csv", "note": "sum over the blocked solver's kernels of one launch"}
        solver_rows.append((case, "all kernels (operands %.3f GB: x%.1f)" % (alg / 1e9, (tot_w + tot_r) / alg), 1, tot_ms, tot_w / 1e9, tot_r / 1e9))
with open(os.path.join(PROF, "r02_solver_summary.csv"), "w") as fh:
    fh.write("case,kernel,calls,avg_ms,pmc_write_GB,pmc_read_GB\n")
    for r_ in solver_rows:
        fh.write("%s,\"%s\",%d,%.3f,%.3f,%.3f\n" % r_)
        print("%-16s %-60s %3d %9.3f ms  write %8.3f GB  read %8.3f GB" % r_)
json.dump(table, open(os.path.join(PROF, "traffic.json"), "w"), indent=1)
with open(os.path.join(PROF, "r02_pmc_traffic.csv"), "w") as fh:
    fh.write("bench,kernel,counter,dispatches,mean_value_KiB\n")
    for r_ in rows:
        fh.write("%s,%s,%s,%d,%.4f\n" % r_)
with open(os.path.join(PROF, "r02_summary.csv"), "w") as fh:
    fh.write("case,kernel,avg_kernel_ms,algorithmic_GB,pmc_traffic_GB,traffic_over_algorithmic,algorithmic_TB_per_s,frac_of_8TBps\n")
    for case, kern, ns, alg, tr in report:
        ms = ns / 1e6 if ns else float("nan")
        fh.write(f"{case},{kern},{ms:.4f},{alg/1e9:.4f},{tr/1e9:.4f},{tr/alg:.4f},{alg/ms/1e9:.3f},{alg/ms/1e9/8:.3f}\n")
        print(f"{case:20s} {kern:22s} {ms:8.4f} ms  alg {alg/1e9:8.3f} GB  pmc {tr/1e9:8.3f} GB ({tr/alg:.4f}x)  {alg/ms/1e9:.2f} TB/s = {alg/ms/1e9/8:.3f}")
