#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.txt
cat gpurun_out/r2a/pytest.txt
timeout 600 python bench.py 2>gpurun_out/r2a/bench.stderr | tail -1 > gpurun_out/r2a/bench.json
cat gpurun_out/r2a/bench.json | python scripts/summ.py
python - <<PY
import json
d = json.load(open("gpurun_out/r2a/bench.json"))
print(json.dumps(d["roofline"], indent=1))
PY
bash scripts/gpu_calib.sh
