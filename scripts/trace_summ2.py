"""Timeline of lv_update from a rocprofv3 kernel trace: per position in the update, kernel, median duration and the median
gap to the previous kernel's end (us).  usage: python scripts/trace_summ2.py <..._kernel_trace.csv> <kernels per update>"""
import csv
import sys
from collections import defaultdict

per = int(sys.argv[2])
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if not any(k in n for k in ("pass_kernel", "search_kernel", "fit_reduce_kernel", "solve_kernel")):
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void lv::", "").replace("lv::", "")[:36]))
rows.sort()
rows = rows[len(rows) % per:]
ups = [rows[i:i + per] for i in range(0, len(rows), per)][-30:]
agg = defaultdict(list)
for seq in ups:
    for j, (s, e, n) in enumerate(seq):
        agg[(j, n)].append((e - s, (s - seq[j - 1][1]) if j else 0))
    agg[(999, "update: first start -> last end")].append((seq[-1][1] - seq[0][0], 0))
med = lambda v: sorted(v)[len(v) // 2] / 1e3
print(f"median over the last {len(ups)} updates: idx kernel dur_us gap_us")
for (j, n), v in sorted(agg.items()):
    print("%3d %-36s %8.1f %6.1f" % (j, n, med([x[0] for x in v]), med([x[1] for x in v])))
