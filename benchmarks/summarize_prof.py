"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel ms per training step."""
import csv, glob, sys
d, nsteps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time per step: {tot / 1e6 / nsteps:.2f} ms")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / nsteps:8.3f} ms/step {float(r['Percentage']):6.2f}%  calls/step "
          f"{int(r['Calls']) / nsteps:7.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
