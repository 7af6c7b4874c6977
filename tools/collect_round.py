"""usage: python tools/collect_round.py round6
Copies the summaries of tools/profile_round.sh from gpurun_out/prof_<round> into profiles/<round>_* and derives
profiles/<round>_fuse_traffic.json (HBM bytes per fuse launch from the PMC passes, corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE is in KB and counts half of the bytes on gfx950 -- checked by the single-tile calibration pass -- WRITE_SIZE in KB)."""
import csv, glob, json, os, re, shutil, collections, sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "round6"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)
DST = os.path.join(ROOT, "profiles")


def find(pattern):
    # gpurun MERGES a run's files into gpurun_out/ (files of earlier runs stay): the newest match is the current run's
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def per_kernel(path, reps):
    """{kernel: (mean counter value, launches per fuse call)}"""
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
        acc[name.split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v) / reps) for k, v in acc.items()}


copies = {
    "bench/**/*kernel_stats.csv": ROUND + "_bench_kernel_stats.csv",
    "bench1/**/*kernel_stats.csv": ROUND + "_bench_kernel_stats_1lane.csv",
    "fuse_launch_windows.csv": ROUND + "_fuse_launch_windows.csv",
    "fuse_variants.txt": ROUND + "_fuse_variants.txt",
    "cb_kstats.txt": ROUND + "_cb_kstats.txt",
    "bench_kstats.txt": ROUND + "_bench_kstats.txt",
    "bench1_kstats.txt": ROUND + "_bench_kstats_1lane.txt",
    "cb_probe.txt": ROUND + "_cb_probe.txt",
    "pair_overhead.txt": ROUND + "_pair_overhead.txt",
    "host_profile.txt": ROUND + "_host_profile.txt",
    "bench_line.json": ROUND + "_bench_line.json",
    "pmc_summary.txt": ROUND + "_pmc_summary.txt",
    "at_size_parity.jsonl": ROUND + "_at_size_parity.jsonl",
    "h2d_probe.txt": ROUND + "_h2d_probe.txt",
    "fuse_classes.txt": ROUND + "_fuse_classes.txt",
    "bench_busy.txt": ROUND + "_bench_busy.txt",
    "lane_gaps.txt": ROUND + "_lane_gaps.txt",
    "launch_rate.txt": ROUND + "_launch_rate.txt",
    "register_phases.txt": ROUND + "_register_phases.txt",
    "fuse_alignment.txt": ROUND + "_fuse_alignment.txt",
}
for pat, name in copies.items():
    f = find(pat)
    if f:
        shutil.copyfile(f, os.path.join(DST, name))
for d in ("int", "frac", "cal", "al128", "cb"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = find(f"pmc_{d}_{c}/**/*counter_collection.csv")
        if f:
            shutil.copyfile(f, os.path.join(DST, f"{ROUND}_pmc_{d}_{c.lower()}.csv"))
log = open(os.path.join(SRC, "bench.log")).read()
m = re.findall(r"^\{.*\}$", log, flags=re.M)
if m:
    open(os.path.join(DST, ROUND + "_bench_line_under_rocprof.json"), "w").write(m[-1] + "\n")

out = {"fetch_correction": 2.0, "source": "profiles/" + ROUND + "_pmc_{int,frac,cal,al128,cb}_{fetch,write}_size.csv (rocprofv3 --pmc, one counter per run, kernel-filtered)"}
cal_f = per_kernel(find("pmc_cal_FETCH_SIZE/**/*counter_collection.csv"), 2)
cal_w = per_kernel(find("pmc_cal_WRITE_SIZE/**/*counter_collection.csv"), 2)
out["calibration"] = {"case": "single 512^3 tile: copy_region_kernel reads exactly what it writes",
                      "copy_fetch_size_kb": cal_f["copy_region_kernel<unsigned short, unsigned short>"][0],
                      "copy_write_size_kb": cal_w["copy_region_kernel<unsigned short, unsigned short>"][0]}
out["calibration"]["fetch_over_write"] = out["calibration"]["copy_fetch_size_kb"] / out["calibration"]["copy_write_size_kb"]
for tag, key, reps in (("int", "integer_offsets", 2), ("frac", "fractional_offsets", 2), ("al128", "aligned_128B_geometry", 2), ("cb", "content_based_probe", 5)):
    f = per_kernel(find(f"pmc_{tag}_FETCH_SIZE/**/*counter_collection.csv"), reps)
    w = per_kernel(find(f"pmc_{tag}_WRITE_SIZE/**/*counter_collection.csv"), reps)
    fk = sum(v * n for v, n in f.values())
    wk = sum(v * n for v, n in w.values())
    out[key] = {"fetch_size_kb": fk, "write_size_kb": wk, "hbm_read_bytes": fk * 1024 * 2.0, "hbm_write_bytes": wk * 1024,
                "hbm_bytes_per_launch": fk * 1024 * 2.0 + wk * 1024,
                "kernels": {k: {"fetch_kb": f[k][0], "write_kb": w.get(k, (0, 0))[0], "launches_per_call": f[k][1]} for k in f}}
out["workload"] = "4x4x4 grid of 512^3 uint16 tiles, 20 % overlap (content_based_probe: 2x2x2 grid of 256^3 tiles, 256^3 chunks + 22 px halo, whole fuse() call)"
out["hbm_bytes_per_launch"] = out["integer_offsets"]["hbm_bytes_per_launch"]
out["note"] = "bench.py's registered mosaic has (recovered) integer offsets, so its 'traffic' is the integer_offsets figure"
import subprocess
sys.path.insert(0, ROOT)
import bench
out["csrc_digest"] = bench.csrc_digest()
try:
    out["git_hash"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    out["git_hash"] = "?"
out["integer_offsets_case"] = "4x4x4 grid of 512^3 uint16 tiles with +-3 px integer jitter (tools/fuse_probe.py 2 2): the geometry of bench.py's registered mosaic"
json.dump(out, open(os.path.join(DST, ROUND + "_fuse_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_launch"] if isinstance(v, dict) and "hbm_bytes_per_launch" in v else None) for k, v in out.items()}, indent=1))
