"""Folds the PMC passes of tools/pmc_traffic.sh into profiles/traffic.json (HBM bytes per launch of the dominant
kernel of each bench) and profiles/r01_pmc_traffic.csv.  WRITE_SIZE and FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte read requests at 64 bytes (checked on a kernel reading a known amount), hence the factor 2."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
KEYS = {"qp_quadrotor": ("qp:quadrotor:16384", "qp_wg_kernel"), "qp_long": ("qp:long:1024", "qp_wg_kernel"),
        "loop_two_mass": ("loop:two_mass:1048576", "closed_loop_lane_kernel"),
        "rollout_msd": ("rollout:msd:1048576", "rollout_lane_kernel"), "gain_long": ("gain:long:1024", "lu_block_kernel")}
table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
rows = []
for name, (key, kern) in KEYS.items():
    vals = {}
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        files = glob.glob(os.path.join(PMC, f"{name}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = []
        for row in csv.DictReader(open(files[0])):
            if kern in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                acc.append(float(row["Counter_Value"]))
        if acc:
            vals[ctr] = sum(acc) / len(acc)
            rows.append((name, kern, ctr, len(acc), vals[ctr]))
    if len(vals) == 2:
        bench = json.loads(open(os.path.join(PMC, f"{name}_WRITE_SIZE.json")).read().strip().splitlines()[-1])
        w, r = vals["WRITE_SIZE"] * 1024, 2 * vals["FETCH_SIZE"] * 1024
        table[key] = {"hbm_bytes_per_launch": w + r, "write_bytes": w, "read_bytes": r,
                      "algorithmic_bytes": bench["roofline"]["bytes_per_unit"] * bench["config"]["batch_per_gpu"],
                      "source": "profiles/r01_pmc_traffic.csv",
                      "note": f"{kern}: WRITE_SIZE x 1024 + 2 x FETCH_SIZE x 1024, mean over its launches"}
json.dump(table, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.csv"), "w") as fh:
    fh.write("bench,kernel,counter,dispatches,mean_value_KiB\n")
    for r_ in rows:
        fh.write("%s,%s,%s,%d,%.4f\n" % r_)
for k, v in table.items():
    print(k, "traffic %.3f GB" % (v["hbm_bytes_per_launch"] / 1e9), "algorithmic %.3f GB" % (v["algorithmic_bytes"] / 1e9))
