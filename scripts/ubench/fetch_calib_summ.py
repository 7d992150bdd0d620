"""FETCH_SIZE (KiB, as rocprofv3 reports it) per kernel of fetch_calib vs the bytes it is known to read."""
import collections
import csv
import glob
import json
import sys

root, known = sys.argv[1], json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            for k in known:
                if k in r["Kernel_Name"]:
                    acc[k].append(float(r["Counter_Value"]))
res = {}
for k, v in sorted(acc.items()):
    rep = sum(v) / len(v) * 1024.0
    res[k] = {"known_bytes": known[k], "fetch_size_bytes": rep, "factor_known_over_reported": known[k] / rep if rep else None, "launches": len(v)}
print(json.dumps(res, indent=1))
