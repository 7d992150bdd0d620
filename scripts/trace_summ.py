"""Timeline of one lv_update from a rocprofv3 kernel trace: kernel, duration, gap to the previous kernel's end.
usage: python scripts/trace_summ.py <..._kernel_trace.csv> [n_updates_from_end]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void lv::", "").replace("lv::", "")[:40]))
rows.sort()
# updates start with kf_begin_kernel
starts = [i for i, r in enumerate(rows) if r[2].startswith("kf_begin")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = defaultdict(list)
for u in range(max(0, len(starts) - 21), len(starts) - 1):
    seq = rows[starts[u]:starts[u + 1]]
    for j, (s, e, n) in enumerate(seq):
        agg[(j, n)].append((e - s, (s - seq[j - 1][1]) if j else 0))
    agg[(999, "update total")].append((seq[-1][1] - seq[0][0], 0))
print("median over the last 20 updates: idx kernel dur_us gap_us")
for (j, n), v in sorted(agg.items()):
    d = sorted(x[0] for x in v)[len(v) // 2] / 1e3
    g = sorted(x[1] for x in v)[len(v) // 2] / 1e3
    print("%3d %-40s %8.1f %6.1f" % (j, n, d, g))
