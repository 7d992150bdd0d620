import collections
import csv
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
for sub in ("sq", "sq2", "fetch", "tcc"):
    f = os.path.join(root, sub, f"{sub}_counter_collection.csv")
    if not os.path.exists(f):
        print("missing", f)
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))  # counter -> pass idx -> values
    order = collections.defaultdict(int)
    seen = {}
    for r in csv.DictReader(open(f)):
        if "search_kernel" not in r["Kernel_Name"]:
            continue
        did = r["Dispatch_Id"]
        if did not in seen:
            seen[did] = order["n"] % 4
            order["n"] += 1
        per[r["Counter_Name"]][seen[did]].append(float(r["Counter_Value"]))
    for c, d in per.items():
        print(sub, c, " ".join("p%d=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(d.items())))
