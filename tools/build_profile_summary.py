"""Per-build kernel times from the rocprofv3 kernel-stats CSV of tools/build_profile.py."""
import csv, sys
path, builds = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = 0.0
for r in rows:
    n = r["Name"]
    name = n.split("namespace)::")[1].split("(")[0] if "namespace)::" in n else n.split("(")[0]
    per = float(r["TotalDurationNs"]) / builds / 1e3
    tot += per
    print(f"{name:28s} calls/build {int(r['Calls']) / builds:6.1f}   avg {float(r['AverageNs']) / 1e3:8.1f} us   per build {per:8.1f} us")
print(f"{'sum':28s} {tot:8.1f} us")
