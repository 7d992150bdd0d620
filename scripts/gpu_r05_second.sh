#!/bin/bash
# round 5, GPU call: the tests touched by the EXT multi-round form / degeneracy mode 1 / the four-rank peer test, the bench line
# with its new legs, and the functional two-rank leg of bench.py on one GPU (rank report in the JSON)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_second
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pass_kernel.py tests/test_gpu_configs.py tests/test_gpu_distributed.py "tests/test_gpu_parity.py::test_degeneracy_hook" -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 900 python bench.py 2>$O/bench.stderr | tail -1 > $O/bench.json
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", d["value"], "pipelined", d["value_pipelined"], "by_value", d["value_by_value"])
print("large_n", d.get("large_n")); print("ext", d.get("ext")); print("parity ok", d["parity"]["ok"])
print("by launch", d["roofline"]["kernel_us_by_launch"], "frac_converged", d["roofline"]["frac_converged"])
PY
timeout 600 python bench.py --gpus 2 --same-device --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench2.stderr | tail -1 > $O/bench_same_device2.json
python - <<PY
import json
d = json.load(open("$O/bench_same_device2.json"))
print("same-device 2 ranks:", d["value"], d["multi_gpu"]["ranks"])
PY
tail -3 $O/bench2.stderr
