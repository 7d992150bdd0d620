"""Print a rocprofv3 kernel_stats.csv compactly: kernel, calls, average / max / total."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    n = r["Name"].split("(")[0][-44:]
    print(f"{n:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  max {float(r['MaxNs'])/1e3:9.1f}  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
