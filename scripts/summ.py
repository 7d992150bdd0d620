import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(round(d["value"]), "it/s | match avg", round(r["avg_kernel_us"], 1), "us | solve", round(r["avg_solve_us"], 1),
          "us | fb", d.get("fallback"), "| per pass", r.get("last_update_match_us_per_pass"), "| frac", round(r["frac"], 4))
