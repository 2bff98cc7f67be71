"""Print the top kernels of a rocprofv3 --stats CSV (`*_kernel_stats.csv`): calls, ms per step, share."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6 / steps:8.2f} ms/step {float(r['Percentage']):5.1f}%")
print("total ms/step", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps)
