#!/bin/bash
# round 5, first GPU call: full GPU suite + smoke, the bench line under the new `value` definition, and the A/B of the headline
# update as two one-step rounds through pass_kernel<.., MULTI> (LV_PASS_SPLIT=1) against the default geometry.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_first
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py 2>$O/bench.stderr | tail -1 > $O/bench_first.json
python scripts/summ.py < $O/bench_first.json
AB="--steps 100 --warmup 10 --regions 5 --no-cpu-baseline --no-cycle --no-large --rotate 0"
for i in 1 2; do
  timeout 300 python bench.py $AB 2>/dev/null | tail -1 > $O/ab_base_$i.json
  LV_PASS_SPLIT=1 timeout 300 python bench.py $AB 2>/dev/null | tail -1 > $O/ab_split_$i.json
done
python - <<PY
import json
for n in ("ab_base_1", "ab_split_1", "ab_base_2", "ab_split_2"):
    try:
        d = json.load(open("$O/%s.json" % n))
        r = d["roofline"]
        print(n, "value %.0f pipelined %.0f by_value %.0f | kernel us by launch %s | phases %s | parity %s" % (
            d["value"], d["value_pipelined"] or 0, d["value_by_value"] or 0, r["kernel_us_by_launch"],
            (r["pass_kernel_phases"] or {}).get("per_launch_us"), d.get("parity", {}).get("ok")))
    except Exception as e:
        print(n, "failed", e)
PY
