"""Print a compact view of a rocprofv3 kernel_stats.csv (name truncated, calls, total ms, avg us, %)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::|ROCPRIM_\d+_NS::|detail::|rocprim::|at::native::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    m = re.match(r"trampoline_kernel<wrapped_(\w+)_config.*?, (\w+)\)::", name)
    print(f"{name[:70]:70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.1f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
