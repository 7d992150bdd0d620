"""Generates tests/golden/stream_oracle.npz: the ORACLE pipeline's trajectory over the 300-update stream of
tests/test_gpu_stream.py (BASELINE configs[4] shape), so that the GPU test need not spend minutes of CPU time on the
GPU box re-running the CPU restatement.  PARITY UNPINNED like everything derived from the oracle (no reference
outputs exist).  Run from the repo root:  python tests/golden/make_golden_stream.py [n_updates]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lvamd  # noqa: E402

lvamd.load()
import lvoracle as oracle  # noqa: E402
import test_gpu_stream as T  # noqa: E402
from limo_velo_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
stream = synth.make_stream(1_048_576, n // 10, n_az=512, map_radius=62.0)
t0 = time.time()
traj, times, sizes, skipped = T.run_stream(T.OracleStream(oracle, stream["map_xyz"], max(8, min(64, os.cpu_count() or 8))), oracle, stream, n,
                                           log=lambda *a: print(a, round(time.time() - t0, 1), flush=True))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stream_oracle.npz"), traj=traj, times=times, sizes=np.array(sizes, np.int64),
                    skipped=skipped, n_map0=len(stream["map_xyz"]), map_checksum=np.float64(stream["map_xyz"].astype(np.float64).sum()))
print("saved", traj.shape, "skipped", skipped)
