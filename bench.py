#!/usr/bin/env python
"""bench.py — KF-update iterations/s of the MI355X-native LIMO-Velo hot path.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run,
one rank per GPU, RCCL; a plain `python bench.py --gpus N` without WORLD_SIZE in the environment
re-launches itself that way on 127.0.0.1).  Prints ONE JSON line on rank 0.

Workload (BASELINE.json metric): 65 536-point scan vs 1 048 576-point map, k = 5, MAX_NUM_ITERS = 3
(4 measurement passes per update), synthetic planar scene (limo-velo_amd/synth.py).
A "step" = one full iterated update (lv_update: up to 4 passes of world transform -> exact 5-NN ->
plane fit -> Jacobian row -> H^T H / H^T h reduction -> 23-dof solve), with map and scan already
resident in HBM.  value = measurement passes per second over the whole job.

N > 1: the scan's points are sharded contiguously across ranks (map replicated); every pass ends in ONE
collective over RCCL/xGMI issued by the library on its stream — the all-gather of every rank's workgroup
partials (one launch per pass; every rank folds them itself), or, if that form fails its start-up self-check,
the all-reduce of the 96-double sums record (three-kernel pass); total work is fixed => "scaling": "strong".
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

M_POINTS = 1_048_576
N_POINTS = 65_536
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def b_alg(m: int) -> int:
    """Algorithmic bytes per point-pass as defined by SURVEY.md §8(d)."""
    return 16 + 16 * math.ceil(math.log2(m / 32)) + 2 * 32 * 16


def pmc_traffic_bytes(kernel="search"):
    """HBM bytes per launch of the dominant kernel (kernel = "pass": lv::pass_kernel, one launch per pass — the single-GPU
    default; "search": lv::search_kernel of the three-kernel pass) from the committed rocprofv3 PMC pass
    of this same command (profiles/pmc_search_*.json, produced by scripts/gpu_profile.sh).  Per
    MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE reports half of
    the bytes of wide coalesced reads, so it is doubled.  The factor is calibrated on known byte counts
    (scripts/ubench/fetch_calib.hip -> profiles/fetch_calib_r02.json): exactly 2.0 for coalesced dword / dwordx3 /
    dwordx4 streams; the search kernel's shape (8-lane groups streaming runs of 56 packed 12-byte records) reads
    1.75x the reported figure in USEFUL bytes, i.e. 2.0x in fetched 128-byte lines (runs start and end mid-line) —
    so 2 x FETCH_SIZE is the HBM/fabric traffic including partial-line over-fetch.  None if no PMC summary is
    committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"pmc_{kernel}_*.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    if "FETCH_SIZE" not in d:
        return None, None
    fetch = d["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * 2.0
    write = d.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) * 1024.0
    return fetch + write, os.path.basename(files[-1])


def rocprof_kernel_avg_us(kernel="lv::pass_kernel<false, false"):
    """Average duration (us) of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/rNN_rocprofv3_kernel_stats.csv, produced by scripts/gpu_profile.sh; the latest round wins) and the file's name.
    (None, None) without one.  Like roofline.traffic it is NOT a measurement of this run: rocprofv3 wraps a process."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_rocprofv3_kernel_stats.csv")))
    if not files:
        return None, None
    for row in csv.DictReader(open(files[-1])):
        if kernel in row["Name"]:
            return float(row["AverageNs"]) / 1e3, os.path.basename(files[-1])
    return None, None


def upd_scan_local(sc, rank: int, world: int):
    from limo_velo_amd.distributed import shard_bounds

    lo, hi = shard_bounds(len(sc["scan_xyz"]), rank, world)
    return sc["scan_xyz"][lo:hi]


def parity_check(g: dict, sc, scan_local, prm, nthreads: int = 16) -> dict:
    """SURVEY 8(d) "parity gates reported with every perf number": the GPU results of THIS run (collected by main(): one
    capturing pass over this rank's points at the initial state, and the timed build's update with its per-pass log) against
    the CPU oracle on the identical inputs — the oracle is the checker here, never the thing measured.  Exact kNN (index
    mismatches, distance bits), valid-mask flips, plane / residual bits, H^T H / H^T h of that pass; then the iterated update:
    pass count, per-pass n_valid, state after every pass, final state and covariance.  The oracle is a restatement of the
    reference whose in-tree half is pinned to the reference's own compiled sources (oracle/_ref; the absent submodules' half is
    not), so "ok" means "equal to the oracle within the stated tolerances"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lvoracle as lo

    tol = {"sums_rel": 1e-10, "state_abs": 1e-9, "P_rel": 1e-9}
    if prm.estimate_extrinsics:   # 12-column solve at this size: condition number 3e6 (tests/test_gpu_configs.py)
        tol = {"sums_rel": 1e-10, "state_abs": 2e-6, "P_rel": 1e-6}
    # the oracle with the configuration the GPU context runs (the lv_params hot keys under the reference's names)
    prm_o = lo.default_params(max_num_iters=prm.MAX_NUM_ITERS, max_dist_plane=prm.MAX_DIST_PLANE, planes_threshold=prm.PLANES_THRESHOLD,
                              estimate_extrinsics=prm.estimate_extrinsics, lidar_noise=prm.LiDAR_noise)
    tree = lo.KdTree(sc["map_xyz"])
    o = lo.iterate(sc["x_init"], sc["map_xyz"], scan_local, params=prm_o, tree=tree, nthreads=nthreads)
    xo, Po, po, tro, so = lo.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], params=prm_o, tree=tree, nthreads=nthreads)
    scale = float(np.abs(o["HTH"]).max())
    g0 = g["g0"]
    out = {
        "against": "oracle/ (CPU restatement of the reference).  Pinned half: world transform, Plane gates, estimate_plane call structure, is_plane, "
                   "Match / chosen set, calculate_H rows = the reference's own sources compiled in place (oracle/_ref, tests/test_oracle_ref.py, "
                   "bit-equal).  Unpinned half (stand-ins on both sides, upstream recall): kNN tie / traversal rule, Eigen QR internals, esekf algebra",
        "points_checked_per_point": int(len(scan_local)),
        "knn_index_mismatches": int((g["idx"] != o["knn_idx"]).any(axis=1).sum()),
        "knn_distance_bit_mismatches": int((g["d2"].view(np.uint32) != o["knn_d2"].view(np.uint32)).any(axis=1).sum()),
        "valid_mask_flips": int((g["valid"] != o["valid"]).sum()),
        "plane_abcd_bit_mismatches": int((g["abcd"].view(np.uint32) != o["abcd"].view(np.uint32)).any(axis=1).sum()),
        "residual_bit_mismatches": int((g["dist"].view(np.uint32) != o["dist"].view(np.uint32)).sum()),
        "max_rel_dHTH": float(np.abs(g0["HTH"] - o["HTH"]).max() / scale),
        "max_abs_dHTh": float(np.abs(g0["HTh"] - o["HTh"]).max()),
        "passes": [int(g["passes"]), int(po)],
        "max_abs_dx": float(np.abs(g["x"] - xo).max()),
        "max_rel_dP": float(np.abs(g["P"] - Po).max() / max(1.0, float(np.abs(Po).max()))),
        "tolerances": tol,
    }
    ok = (out["knn_index_mismatches"] == 0 and out["knn_distance_bit_mismatches"] == 0 and out["valid_mask_flips"] == 0
          and out["plane_abcd_bit_mismatches"] == 0 and out["residual_bit_mismatches"] == 0
          and out["max_rel_dHTH"] <= tol["sums_rel"] and out["max_abs_dHTh"] <= tol["sums_rel"] * max(1.0, float(np.abs(o["HTh"]).max()))
          and out["passes"][0] == out["passes"][1] and out["max_abs_dx"] < tol["state_abs"] and out["max_rel_dP"] < tol["P_rel"])
    if g.get("x_res") is not None:
        # the timed (device-resident, pipelined) steps ran the same launches as the update by value that is checked here: the
        # posterior they left on the device must be that update's, bit for bit
        out["resident_step_equals_update_by_value"] = bool(np.array_equal(g["x_res"], g["x"]) and np.array_equal(g["P_res"], g["P"]))
        ok = ok and out["resident_step_equals_update_by_value"]
    if g.get("sums") is not None:
        np_ = min(int(g["passes"]), int(po))
        out["n_valid_per_pass"] = [[int(v["n_valid"]) for v in g["sums"]], [int(v["n_valid"]) for v in so]]
        out["max_abs_dstate_per_pass"] = [float(np.abs(g["tr"][i] - tro[i]).max()) for i in range(np_)]
        ok = ok and out["n_valid_per_pass"][0] == out["n_valid_per_pass"][1] and all(v < tol["state_abs"] for v in out["max_abs_dstate_per_pass"])
    out["ok"] = bool(ok)
    return out


def cycle_64k(ctx, sc, capi, reps: int = 10) -> dict:
    """One whole cycle of the reference's loop (src/main.cpp:75-103) at the headline size, device-resident stages timed with
    the PCIe legs beside them: a 65 536-point hesai PointCloud2 message in (lv_cloud_ingest: H2D + decode + time sort) ->
    de-skew of the buffered window + Morton order (lv_scan_deskew_window, no voxel grid so the scan keeps its 64k points) ->
    the iterated update on the resident filter (lv_correct) -> map insert with ikd-Tree down-sampling (lv_map_add_scan) ->
    lv_cloud_clear.  The sensor is at rest (identity de-skew), so every cycle sees the benchmark's scan."""
    import struct  # noqa: F401

    n = len(sc["scan_xyz"])
    rec = np.zeros(n, np.dtype({"names": ["x", "y", "z", "intensity", "timestamp", "ring"], "formats": ["<f4", "<f4", "<f4", "u1", "<f8", "<u2"],
                                "offsets": [0, 4, 8, 16, 24, 32], "itemsize": 48}))
    rec["x"], rec["y"], rec["z"] = sc["scan_xyz"].T
    t0 = 100.0
    rec["timestamp"] = t0 + np.arange(n) * (0.1 / n)
    raw = rec.tobytes()
    fmt = ctx.cloud_format_preset(capi.LIDAR_HESAI)
    prm = capi.IngestParams(int(t0 * 1e6), 0, 0, 0.1, 1, 0.0)   # stamp, not real time, stamp at the end, rotation time, every point, min_dist 0
    states = np.concatenate([capi.motion_state(time=t0 - 0.01), capi.motion_state(time=t0 + 0.11)])
    xt2 = capi.motion_state(time=t0 + 0.1)
    m0 = ctx.map_size()

    def one(timed):
        ts = []
        ctx.synchronize(); a = time.perf_counter()
        ctx.cloud_ingest(raw, n, fmt, prm); ctx.synchronize(); b = time.perf_counter()
        nw = ctx.scan_deskew_window(t0 - 1.0, t0 + 1.0, states, xt2, 0.0); ctx.synchronize(); c = time.perf_counter()
        ctx.filter_set(sc["x_init"], sc["P0"])
        ctx.correct(want_passes=False); ctx.synchronize(); d = time.perf_counter()
        ctx.map_add_scan(True); ctx.synchronize(); e = time.perf_counter()
        ctx.cloud_clear(1e300); ctx.synchronize(); f = time.perf_counter()
        return nw, [b - a, c - b, d - c, e - d, f - e]

    for _ in range(3):
        nw, _ = one(False)
    acc = np.zeros(5)
    for _ in range(reps):
        nw, t = one(True)
        acc += t
    acc = acc / reps * 1e3
    # without a synchronisation between the stages (what a host program does): the device-resident part back to back
    ctx.synchronize(); a = time.perf_counter()
    for _ in range(reps):
        ctx.cloud_ingest(raw, n, fmt, prm)
        ctx.scan_deskew_window(t0 - 1.0, t0 + 1.0, states, xt2, 0.0)
        ctx.filter_set(sc["x_init"], sc["P0"])
        ctx.correct(want_passes=False)
        ctx.map_add_scan(True)
        ctx.cloud_clear(1e300)
    ctx.synchronize()
    chained = (time.perf_counter() - a) / reps * 1e3
    out = {"points_in_window": int(nw), "ingest_pcie_ms": round(float(acc[0]), 4), "deskew_sort_ms": round(float(acc[1]), 4),
           "correct_ms": round(float(acc[2]), 4), "map_insert_ms": round(float(acc[3]), 4), "clear_ms": round(float(acc[4]), 4),
           "device_resident_ms": round(float(acc[1] + acc[2] + acc[3]), 4), "pcie_inclusive_ms": round(float(acc.sum()), 4),
           "pcie_inclusive_chained_ms": round(float(chained), 4),
           "map_points_before_after": [int(m0), int(ctx.map_size())],
           "note": "stage times with a synchronisation after every stage; chained = the same calls back to back"}
    return out


def predicted_scaling(world: int, n_points: int, passes: int = 4, collective_us: float = 6.0) -> dict | None:
    """The builder's PREDICTION for an N-rank run, so that the first execution across xGMI is read against it:
    one rank's update on its shard from the newest committed single-GPU sweep (profiles/shard_size_sweep_r*.txt, by-value
    one-launch form, log-interpolated between the measured sizes) + `passes` collectives of ~6 us each (4.6 us all-gather measured
    with ONE rank + a launch boundary: a real ring over xGMI will cost more, so the ratio is an upper bound)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "shard_size_sweep_r*.txt")))
    if not files:
        return None
    pts = []
    for line in open(files[-1]):
        mm = re.search(r"fused scan\s+(\d+) points:\s+([0-9.]+) us per update", line)
        if mm:
            pts.append((int(mm.group(1)), float(mm.group(2))))
    if len(pts) < 2:
        return None
    pts.sort()
    xs, ys = np.log(np.array([p[0] for p in pts], float)), np.array([p[1] for p in pts], float)

    def us_of(n):
        return float(np.interp(np.log(max(n, pts[0][0])), xs, ys))
    one = us_of(n_points)
    shard = -(-n_points // world)
    pred = us_of(shard) + (passes * collective_us if world > 1 else 0.0)
    return {"source": "profiles/" + os.path.basename(files[-1]), "shard_points": int(shard), "us_per_update_1gpu": round(one, 1),
            "us_per_update_predicted": round(pred, 1), "ratio_vs_1gpu_predicted": round(one / pred, 3),
            "collective_us_assumed": collective_us,
            "note": "a prediction from single-GPU sweeps, NOT a measurement: nothing has crossed xGMI in rounds 1-6"}


def cpu_baseline(sc, passes_expected: int, budget_s: float = 20.0) -> dict:
    """Times the CPU oracle (a port/restatement — the reference binary cannot be built here) on this
    host's cores, same scene, same update, bounded to ~budget_s seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lvoracle as lo

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    tree = lo.KdTree(sc["map_xyz"])
    args = (sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"])

    def rate(threads, seconds):
        lo.update(*args, tree=tree, nthreads=threads)  # warm-up
        t0, passes, reps = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < seconds and reps < 40:
            passes += lo.update(*args, tree=tree, nthreads=threads)[2]
            reps += 1
        return passes / (time.perf_counter() - t0), reps, passes

    # the reference matches on MP_PROC_NUM = 3 OpenMP threads (CMakeLists.txt:23-26); also find this
    # host's best thread count (over-subscription hurts: the sweep picks the fastest)
    ref3 = rate(min(3, ncpu), budget_s * 0.15)[0]
    cand = sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu} | {min(ncpu, 8)})
    sweep = {t: rate(t, budget_s * 0.08)[0] for t in cand}
    best_t = max(sweep, key=sweep.get)
    best, reps, passes = rate(best_t, budget_s * 0.4)
    ref_build = reference_build_baseline(sc)
    return {
        "value": best,
        "unit": "KF-update iters/s",
        "cores": best_t,
        "kind": "port",
        # the reference's Localizator::correct call chain (src/Modules/{Localizator,Mapper}.cpp, src/Objects/*.cpp, src/Utils/Utils.cpp
        # compiled in place: oracle/_ref) over STAND-INS for its un-vendored dependencies (kNN = the oracle's kd-tree, not ikd-Tree;
        # filter algebra = the oracle's, not IKFoM's: hence kind "reference_callers+standins") on one core, where the prebuilt library travelled with the snapshot; `value` stays the
        # faster multi-threaded port (the conservative comparison)
        "reference_build": ref_build,
        "sample": f"{reps} full updates ({passes} passes) of the same 64k-vs-1M workload on the oracle (pointer kd-tree, "
                  f"OpenMP {best_t} threads = fastest of {sorted(sweep)} on a {ncpu}-cpu host); reference configuration "
                  f"MP_PROC_NUM=3 threads: {ref3:.1f} iters/s",
        "reference_config_3_threads": ref3,
    }


def reference_build_baseline(sc, max_updates: int = 2):
    """The reference's own code on the CPU: oracle/_ref/liblvref.so = /root/reference/src compiled in place against stand-in headers
    (oracle/ref_build; kNN = the oracle's pointer kd-tree behind ikd-Tree's interface, the filter algebra = the oracle's behind
    esekf's; Mapper::match's loop runs on ONE thread: its OpenMP form races on push_back).  A bounded sample of the same
    64k-vs-1M update.  None when the library did not travel (it is built only where the reference is mounted)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import lvref

        if not os.path.exists(lvref._LIB_PATH):
            return None
        lvref.set_config()
        lvref.reset()
        lvref.map_add(sc["map_xyz"])
        t0, passes, reps = time.perf_counter(), 0, 0
        while reps < max_updates and time.perf_counter() - t0 < 12.0:
            passes += lvref.update(sc["x_init"], sc["P0"], sc["scan_xyz"])[2]
            reps += 1
        dt = time.perf_counter() - t0
        lvref.reset()
        return {"value": passes / dt, "unit": "KF-update iters/s", "cores": 1, "kind": "reference_callers+standins",
                "sample": f"{reps} updates ({passes} passes) of the same workload through Localizator::correct of the reference's compiled "
                          "sources (first one includes the stand-in tree's build); stand-ins: kNN, esekf algebra"}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def self_launch(n: int) -> int:
    """Re-executes this script under torch.distributed.run with n ranks on 127.0.0.1 (free port); the children
    inherit stdout, so rank 0's JSON line is the only line printed."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--regions", type=int, default=9, help="timed regions of exactly --steps steps each (each bracketed by barrier + "
                    "synchronise, MAX over ranks); value = the median region, the others give value_min / value_max")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="override lanes_per_query")
    ap.add_argument("--voxel", type=float, default=0.0, help="override voxel_size")
    ap.add_argument("--extrinsics", action="store_true", help="estimate_extrinsics = true (12 live Jacobian columns; not the headline config)")
    ap.add_argument("--rotate", type=int, default=16, help="cold-cache leg (N=1): cycle this many distinct scans / poses of the same map (their "
                    "neighbourhood buckets together exceed the 256 MB Infinity Cache); 0 = skip")
    ap.add_argument("--no-phases", action="store_true", help="skip the in-kernel phase stamps leg (N=1, one launch per pass)")
    ap.add_argument("--force-comm", action="store_true", help="diagnostic: take the multi-GPU route (library RCCL) even at N=1")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gate (GPU results of this run vs the CPU oracle)")
    ap.add_argument("--no-ext", action="store_true", help="skip the estimate_extrinsics leg (config/xaloc.yaml's setting at the headline sizes)")
    ap.add_argument("--no-large", action="store_true", help="skip the large-N leg (262 144-point scan vs 5 M-point map on one GPU)")
    ap.add_argument("--no-cfgs", action="store_true", help="skip the BASELINE configs[1] / configs[2] legs (ring scans vs 500k / 2M-point maps)")
    ap.add_argument("--no-cycle", action="store_true", help="skip the whole-cycle leg (message in -> de-skew -> correct -> map insert)")
    ap.add_argument("--resident-only", action="store_true", help="profiling aid (scripts/gpu_profile.sh): only the timed resident steps — no "
                    "by-value legs, no event-instrumented repetition — so that a rocprofv3 run of this command sees the timed step alone")
    ap.add_argument("--same-device", action="store_true", help="FUNCTIONAL leg, not a measurement of scaling: the N ranks all use GPU 0 and "
                    "exchange their partials through the peer-mapped buffers (HIP IPC; RCCL refuses two ranks on one device) — exercises "
                    "the self-launch, sharding, forms and watchdog logic of an N > 1 run on a one-GPU box")
    args = ap.parse_args()

    import torch

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # a multi-rank run that stops making progress (a rank lost inside a collective) must end with a traceback, not hang
        # until somebody else's limit kills it
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ.get("LV_BENCH_WATCHDOG_S", "900")), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL) and relay
        # rank 0's single JSON line; under torch.distributed.run the environment is already there
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # stdout carries exactly ONE line (the JSON record of rank 0): everything else that ends up on file descriptor 1 — RCCL's
    # version banner at communicator teardown, library chatter — is sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    same_dev = bool(args.same_device) and world > 1
    if same_dev:
        local_rank = 0                      # every rank on GPU 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if same_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only: the data path is the peer exchange
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def reduce_over_ranks(values, op):
        """MAX / MIN of a list of numbers over the ranks (device tensors with RCCL, host tensors with gloo)."""
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if same_dev else "cuda")
        dist.all_reduce(t, op=op)
        return [float(v) for v in t.tolist()]

    import lvamd

    lvamd.load()
    from limo_velo_amd import capi, synth
    from limo_velo_amd.distributed import HipEngine, ShardedUpdater, init_library_comm

    sc = synth.make_scene(M_POINTS, N_POINTS)
    kw = {}
    if args.lanes:
        kw["lanes_per_query"] = args.lanes
    if args.voxel:
        kw["voxel_size"] = args.voxel
    if args.extrinsics:
        kw["estimate_extrinsics"] = 1
    prm = capi.default_params(**kw)
    ctx = capi.Context(prm, device=local_rank)
    ctx.map_build(sc["map_xyz"])
    collective = "none"
    engine = ctx
    if same_dev:
        from limo_velo_amd.distributed import init_peer_gather

        init_peer_gather(ctx, dist, rank, world)
        engine = HipEngine(ctx, torch, multi=False, library_comm=True)
        collective = "peer-mapped exchange (HIP IPC), all ranks on ONE GPU [functional leg]"
    elif world > 1:
        # preferred: RCCL issued by the library itself on its stream (no host round trip per pass); if the
        # communicator cannot be created, the same all-reduce goes through torch.distributed pass by pass
        try:
            init_library_comm(ctx, dist, torch, rank, world)
            engine = HipEngine(ctx, torch, multi=False, library_comm=True)
            collective = "rccl (library, on the context stream)"
        except Exception as e:  # noqa: BLE001
            print(f"[bench] library RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
            collective = "rccl (torch.distributed, per pass from the host)"
        all_have = int(reduce_over_ranks([1 if collective.startswith("rccl (library") else 0], dist.ReduceOp.MIN)[0])
        if all_have == 0 and engine is not ctx:   # not every rank got a communicator: all fall back together
            ctx.comm_destroy()
            engine = ctx
            collective = "rccl (torch.distributed, per pass from the host)"
    if world == 1 and args.force_comm:
        init_library_comm(ctx, None, torch, 0, 1)
        collective = "rccl (library, on the context stream) [forced, 1 rank]"
    upd = ShardedUpdater(engine, rank, world, dist, torch)
    upd.scan_set(sc["scan_xyz"])
    n_local = upd.n_local
    if same_dev:
        from limo_velo_amd.distributed import shard_bounds

        lo0, hi0 = shard_bounds(N_POINTS, 0, world)
        ctx.comm_set_shard_max(hi0 - lo0)
    if collective.startswith("rccl (library"):
        # With the library's communicator a pass is ONE launch + ONE collective: every rank's workgroup partials are
        # all-gathered (in place, on the context stream) and every rank's next launch folds them itself.  Checked here, on
        # the ranks this run really has, against the three-kernel form (search / fit / reduce -> all-reduce -> solve): any
        # rank that fails, or disagrees, sends every rank back to the three-kernel form.
        from limo_velo_amd.distributed import shard_bounds

        lo0, hi0 = shard_bounds(N_POINTS, 0, world)
        ok = 1
        try:
            ctx.comm_set_shard_max(hi0 - lo0)
            xa, Pa, pa = upd.update(sc["x_init"], sc["P0"])
            took = ctx.last_update_fused()
            ctx.set_comm_fused(False)
            xb, Pb, pb = upd.update(sc["x_init"], sc["P0"])
            ctx.set_comm_fused(True)
            ok = int(bool(took) and pa == pb and np.abs(xa - xb).max() < 1e-9 and np.abs(Pa - Pb).max() < 1e-9)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] all-gather form failed on rank {rank}: {e}", file=sys.stderr)
            ok = 0
        if dist is not None:
            ok = int(reduce_over_ranks([ok], dist.ReduceOp.MIN)[0])
        if ok:
            collective += ": one launch per pass, ncclAllGather of the workgroup partials"
        else:
            ctx.set_comm_fused(False)
            collective += ": three-kernel pass, ncclAllReduce of the 96-double record (the all-gather form failed its self-check)"

    lib_comm_early = collective.startswith("rccl (library") or same_dev

    def barrier_sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            if same_dev:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        x, P, passes = upd.update(sc["x_init"], sc["P0"])

    # ---- what a step is.  The reference's loop reads the posterior after every correction (src/main.cpp:84-86: `correct` ->
    # `latest_state`), so a STEP = "set the prior, run the iterated update, read the posterior back" with one host wait per step:
    # lv_filter_set + lv_correct + lv_filter_get on the device-resident filter (row f-3; the state rides in the first launch's
    # kernel arguments, the posterior comes back through the pinned mailbox).  Every step starts from the same prior, so every
    # step is the same four passes.  That is `value` (round 5; rounds 1-3 timed lv_update by value — the same launches with x / P
    # handed over as host buffers: `value_by_value_sync`, a top-level sibling; round 4 reported the PIPELINED rate as `value` —
    # K steps enqueued back to back with no posterior read in between, which no caller of the reference's API can do: it is kept
    # as `value_pipelined`).  Routes without the library's own collective (torch.distributed driving the all-reduce pass by pass)
    # have no resident form: their step is the by-value update.
    import ctypes as C

    resident = (world == 1 or lib_comm_early) and not os.environ.get("LV_BENCH_BY_VALUE")
    x0c = np.ascontiguousarray(sc["x_init"], np.float64)
    P0c = np.ascontiguousarray(sc["P0"], np.float64)
    x0p, P0p = x0c.ctypes.data_as(C.c_void_p), P0c.ctypes.data_as(C.c_void_p)

    xg_buf, Pg_buf = np.zeros(26), np.zeros(23 * 23)
    xgp, Pgp = xg_buf.ctypes.data_as(C.c_void_p), Pg_buf.ctypes.data_as(C.c_void_p)

    def timed_regions(n_regions, form):
        """n_regions regions of exactly K steps each, no instrumentation on the stream, every region bracketed by barrier +
        synchronise on both sides (a region of 20 steps is 2.7 ms: a single one is at the mercy of whatever else the box does in
        those milliseconds, so the median region is reported and the line carries the rest).  form: "sync" = resident filter, the
        posterior read back every step; "pipelined" = resident filter, steps enqueued back to back; "by_value" = lv_update.
        Returns (dt, passes) per region."""
        dts, ps = [], []
        lib, h = ctx.lib, ctx.h
        for _ in range(max(n_regions, 1)):
            tp = 0
            barrier_sync()
            t0 = time.perf_counter()
            if form == "by_value":
                for _ in range(args.steps):
                    tp += upd.update(sc["x_init"], sc["P0"])[2]
            elif form == "sync":
                for _ in range(args.steps):
                    if lib.lv_filter_set(h, x0p, P0p) or lib.lv_correct(h, None) or lib.lv_filter_get(h, xgp, Pgp):
                        raise RuntimeError(lib.lv_last_error().decode())
            else:
                for _ in range(args.steps):
                    if lib.lv_filter_set(h, x0p, P0p) or lib.lv_correct(h, None):
                        raise RuntimeError(lib.lv_last_error().decode())
            barrier_sync()
            dts.append(time.perf_counter() - t0)
            if form != "by_value":
                tp = ctx.last_passes() * args.steps      # (every step is the same update: the last one's pass count)
            ps.append(tp)
        if dist is not None:   # a region lasts as long as its slowest rank
            dts = reduce_over_ranks(dts, dist.ReduceOp.MAX)
        return dts, ps

    def median_region(dts, ps):
        order = sorted(range(len(dts)), key=lambda i: ps[i] / dts[i])
        return order[(len(order) - 1) // 2]      # (the lower of the two middle ones for an even count)

    if resident:
        for _ in range(max(args.warmup // 2, 2)):
            ctx.lib.lv_filter_set(ctx.h, x0p, P0p)
            ctx.lib.lv_correct(ctx.h, None)
        ctx.synchronize()
    region_dt, region_passes = timed_regions(args.regions, "sync" if resident else "by_value")
    mid = median_region(region_dt, region_passes)
    dt, total_passes = region_dt[mid], region_passes[mid]
    by_value_sync = pipelined = None
    x_res = P_res = None
    if resident:
        x_res, P_res = xg_buf.copy(), Pg_buf.reshape(23, 23).copy()     # the posterior the last timed step read back

        def side_leg(form, note):
            d_, p_ = timed_regions(min(args.regions, 3) if not args.resident_only else 1, form)
            m_ = median_region(d_, p_)
            return {"value": p_[m_] / d_[m_], "ms_per_step": d_[m_] / args.steps * 1e3,
                    "value_per_region": [round(p / d, 1) for p, d in zip(p_, d_)], "note": note}

        pipelined = side_leg("pipelined", "lv_filter_set + lv_correct enqueued back to back, one synchronisation per region, no posterior "
                             "read between steps (round 4's definition of `value`; not a call pattern the reference's loop has)")
        if not args.resident_only:
            by_value_sync = side_leg("by_value", "lv_update by value, one host round trip per step (the definition of `value` in rounds 1-3)")
    x, P, passes = upd.update(sc["x_init"], sc["P0"])
    # ---- the same K steps again with HIP events around the dominant kernel (ctx stream) --------------
    # (event records between kernels add ~5 us gaps each, so they are kept out of the timed region; kernel
    # durations themselves are unaffected and must agree with the rocprofv3 summary under profiles/)
    lib_comm = collective.startswith("rccl (library") or same_dev   # (lv_update itself runs the passes and the exchange)

    def events_leg(steps):
        """steps profiled updates: average device time of the dominant kernel, of the rest of a pass, and — with a library
        communicator — of the pass' collective (HIP events on the context stream around each)."""
        ctx.set_profiling(True)
        k_ms, s_ms, cnt, c_us = 0.0, 0.0, 0, np.zeros(8)
        for _ in range(steps):
            _, _, p = upd.update(sc["x_init"], sc["P0"])
            tm = ctx.timing()
            k_ms += tm["last_reduce_ms"] * p
            s_ms += tm["last_solve_ms"] * p
            c_us[:p] += np.array(tm["pass_collective_ms"][:p]) * 1e3
            per_launch_us[:p] += np.array(tm["pass_match_ms"][:p]) * 1e3
            per_launch_n[:p] += 1
            cnt += p
        ctx.set_profiling(False)
        return k_ms, s_ms, cnt, (c_us / max(steps, 1))

    kern_ms, kern_cnt, solve_ms, coll_us = 0.0, 0, 0.0, np.zeros(8)
    per_launch_us, per_launch_n = np.zeros(8), np.zeros(8)     # the dominant kernel's HIP-event time by launch index within an update
    if (world == 1 or lib_comm) and not args.resident_only:   # lv_update itself runs the passes: per-kernel events exist
        kern_ms, solve_ms, kern_cnt, coll_us = events_leg(args.steps)
    fused_main = bool(ctx.last_update_fused())
    # ---- N > 1: the OTHER form of the multi-GPU pass, the same K steps (timed like the headline region, then with events), so
    # that the first run on a node yields the breakdown of both: "allgather_one_launch" = one launch per pass + ncclAllGather
    # of the workgroup partials, "allreduce_three_kernel" = search / fit / reduce -> ncclAllReduce of 768 bytes -> solve
    forms = None
    # ---- N > 1: what every rank ended with, gathered so that the line can say so (a multi-GPU run must not be able to fail
    # silently): the communicator's size as the LIBRARY sees it, every rank's shard, and whether all ranks hold the same bits
    rank_report = None
    if world > 1:
        objs = [None] * world
        try:
            nranks_lib = int(ctx.comm_world())
        except Exception:  # noqa: BLE001
            nranks_lib = None
        dist.all_gather_object(objs, (rank, int(n_local), nranks_lib, x.tobytes(), P.tobytes(), int(passes), bool(ctx.last_update_fused())))
        rank_report = {"world": world, "library_comm_nranks": [o[2] for o in objs], "points_per_rank": [o[1] for o in objs],
                       "passes_per_rank": [o[5] for o in objs], "one_launch_per_pass_per_rank": [o[6] for o in objs],
                       "ranks_bitwise_equal": bool(all(o[3] == objs[0][3] and o[4] == objs[0][4] for o in objs)),
                       "points_total": int(sum(o[1] for o in objs))}
    if same_dev:
        # the one form this leg has: the peer-mapped one-launch pass; ranks must end bitwise equal
        objs = [None] * world
        dist.all_gather_object(objs, (x.tobytes(), P.tobytes()))
        forms = {"peer_mapped_one_launch": {"avg_kernel_us": kern_ms / max(kern_cnt, 1) * 1e3,
                                             "exchange_kernel_us_per_pass": [round(float(v), 2) for v in coll_us[:4]],
                                             "ranks_bitwise_equal": bool(all(o == objs[0] for o in objs)),
                                             "one_launch_per_pass": bool(fused_main)}}
    elif lib_comm and (world > 1 or args.force_comm):   # (--force-comm: the same legs with one rank, to exercise this code on one GPU)
        def timed_form():
            barrier_sync()
            a, tp = time.perf_counter(), 0
            for _ in range(args.steps):
                tp += upd.update(sc["x_init"], sc["P0"])[2]
            barrier_sync()
            d = time.perf_counter() - a
            if dist is not None:
                d = reduce_over_ranks([d], dist.ReduceOp.MAX)[0]
            return tp / d, d / args.steps * 1e3

        def describe(rate_ms, ev):
            k_ms, s_ms, cnt, c_us = ev
            return {"iters_per_s": rate_ms[0], "ms_per_step": rate_ms[1], "avg_kernel_us": k_ms / max(cnt, 1) * 1e3,
                    "avg_rest_of_pass_us": s_ms / max(cnt, 1) * 1e3, "collective_us_per_pass": [round(float(v), 2) for v in c_us[:4]]}

        main_name = "allgather_one_launch" if fused_main else "allreduce_three_kernel"
        forms = {main_name: describe((None, None), (kern_ms, solve_ms, kern_cnt, coll_us))}   # (its rate = the headline value, filled in below)
        if os.environ.get("LV_BENCH_PEER", "0") != "0" and dist is not None:
            # (OPT-IN since round 5, LV_BENCH_PEER=1: this leg has never run across GPUs, and a fault in it — a bad peer mapping is a
            # GPU page fault, not an exception — would take the whole line with it, the two RCCL forms included; it has not run
            # across GPUs yet: a failure stays inside this try and inside the second context, a lost rank ends a wait after
            # LV_PEER_TIMEOUT_MS) the same one-launch form with the partials pulled out of peer-mapped buffers
            # (lv_comm_peer_export / _init) by a second context per rank
            try:
                from limo_velo_amd.distributed import init_peer_gather

                with capi.Context(prm, device=local_rank) as c2:
                    c2.map_build(sc["map_xyz"])
                    init_peer_gather(c2, dist, rank, world)
                    u2 = ShardedUpdater(HipEngine(c2, torch, multi=False, library_comm=True), rank, world, dist, torch)
                    u2.scan_set(sc["scan_xyz"])
                    for _ in range(max(args.warmup // 2, 2)):
                        x2, P2, p2 = u2.update(sc["x_init"], sc["P0"])
                    barrier_sync(); c2.synchronize()
                    a2, tp2 = time.perf_counter(), 0
                    for _ in range(args.steps):
                        tp2 += u2.update(sc["x_init"], sc["P0"])[2]
                    c2.synchronize(); barrier_sync()
                    d2 = time.perf_counter() - a2
                    d2 = reduce_over_ranks([d2], dist.ReduceOp.MAX)[0]
                    forms["peer_mapped_one_launch"] = {"iters_per_s": tp2 / d2, "ms_per_step": d2 / args.steps * 1e3,
                                                       "agrees_with_headline_form": bool(np.abs(x2 - x).max() < 1e-9)}
            except Exception as e:  # noqa: BLE001
                forms["peer_mapped_one_launch"] = {"error": str(e)}
        if fused_main:   # the headline ran the all-gather form: also time the all-reduce form
            ctx.set_comm_fused(False)
            for _ in range(max(args.warmup // 2, 2)):
                upd.update(sc["x_init"], sc["P0"])
            r = timed_form()
            forms["allreduce_three_kernel"] = describe(r, events_leg(args.steps))
            ctx.set_comm_fused(True)
            upd.update(sc["x_init"], sc["P0"])
    # ---- cold-cache leg: the headline loop replays ONE scan, whose ~45 MB of buckets stay in the 256 MB Infinity
    # Cache from step to step; here K scans from K poses of the same map are cycled (lv_scan_set + lv_update each),
    # so the buckets a scan touches have been evicted since its previous turn.  Only the search kernel's HIP-event
    # time is used (lv_scan_set is host work between the updates, outside the events).
    cold = None
    if world == 1 and args.rotate >= 2:
        extra = [synth.make_extra_scan(M_POINTS, N_POINTS, k) for k in range(args.rotate)]
        ctx.set_profiling(True)
        c_ms, c_cnt, c_first, c_rest, c_upd = 0.0, 0, 0.0, 0.0, 0
        for rnd in range(3):          # round 0 untimed (first touch of every page)
            for e in extra:
                ctx.scan_set(e["scan_xyz"])
                _, _, p, _, _ = ctx.update(e["x_init"], sc["P0"], want_trace=False)
                if rnd:
                    tm = ctx.timing()
                    c_ms += tm["last_reduce_ms"] * p
                    c_cnt += p
                    c_first += tm["pass_match_ms"][0]
                    c_rest += sum(tm["pass_match_ms"][1:p])
                    c_upd += 1
        ctx.set_profiling(False)
        upd.scan_set(sc["scan_xyz"])
        cold_s = c_ms / max(c_cnt, 1) * 1e-3
        cold = {"scans_cycled": args.rotate, "avg_kernel_us": cold_s * 1e6,
                "achieved": b_alg(M_POINTS) * N_POINTS / cold_s / 1e9 if cold_s > 0 else 0.0,
                "first_pass_us": c_first / max(c_upd, 1) * 1e3, "later_pass_us": c_rest / max(c_cnt - c_upd, 1) * 1e3}
        cold["frac"] = cold["achieved"] / HBM_PEAK_GBS
    # ---- inside the dominant kernel (one launch per pass only): wall-clock stamps of pass_kernel's phases, taken by a
    # SECOND context created with LV_PASS_CLK=1 (the timed context above carries no instrumentation): how long the search
    # phase of a launch lasts (prologue end -> the workgroup's search barrier, median over the workgroups), per launch
    phases = None
    if world == 1 and ctx.last_update_fused() and not args.no_phases:
        os.environ["LV_PASS_CLK"] = "1"
        try:
            with capi.Context(prm, device=local_rank) as c2:
                c2.map_build(sc["map_xyz"])
                c2.scan_set(sc["scan_xyz"])
                acc = []
                for i in range(25):
                    c2.update(sc["x_init"], sc["P0"], want_trace=False)
                    if i >= 5:
                        clk, nwg_c = c2.pass_clocks()
                        w = clk[:, :nwg_c, 16:].astype(np.float64) / 100.0          # wall clock, us (100 MHz)
                        acc.append([[np.median(w[li, :, 3] - w[li, :, 0]), np.median(w[li, :, 6] - w[li, :, 3]),
                                     np.median(w[li, :, 9] - w[li, :, 6]), w[li, :, 9].max() - w[li, :, 0].min()]
                                    for li in range(clk.shape[0] - 1)])
                m = np.median(np.array(acc), axis=0)
                phases = {"per_launch_us": {"prologue": [round(v, 2) for v in m[:, 0]], "search": [round(v, 2) for v in m[:, 1]],
                                            "fit_and_partial": [round(v, 2) for v in m[:, 2]], "span": [round(v, 2) for v in m[:, 3]]},
                          "note": "medians over the workgroups (span: first start -> last search/fit end) and over 20 updates; wall clock inside the kernel"}
        finally:
            del os.environ["LV_PASS_CLK"]
    # ---- the large-N figure, driver-timed (N=1 only): BASELINE configs[3]'s sizes on ONE GPU — 262 144-point scan against a
    # 5 242 880-point map (B_alg 1328 B per point-pass), same synchronised step as `value`, in a context of its own
    large = None
    if world == 1 and not args.no_large and not args.resident_only:
        try:
            ML, NL = 5 * 1_048_576, 262_144
            scl = synth.make_scene(ML, NL)
            with capi.Context(prm, device=local_rank) as c3:
                c3.map_build(scl["map_xyz"])
                c3.scan_set(scl["scan_xyz"])
                xl = np.ascontiguousarray(scl["x_init"], np.float64); Pl = np.ascontiguousarray(scl["P0"], np.float64)
                xlp, Plp = xl.ctypes.data_as(C.c_void_p), Pl.ctypes.data_as(C.c_void_p)
                for _ in range(5):
                    c3.filter_set(xl, Pl); c3.correct(want_passes=False); c3.filter_get()
                rates = []
                for _ in range(3):
                    c3.synchronize(); t0 = time.perf_counter()
                    for _ in range(max(args.steps // 4, 5)):
                        if c3.lib.lv_filter_set(c3.h, xlp, Plp) or c3.lib.lv_correct(c3.h, None) or c3.lib.lv_filter_get(c3.h, xgp, Pgp):
                            raise RuntimeError(c3.lib.lv_last_error().decode())
                    c3.synchronize()
                    rates.append((time.perf_counter() - t0) / max(args.steps // 4, 5))
                pl_ = c3.last_passes()
                us = sorted(rates)[1] * 1e6
                large = {"workload": f"{NL}-pt scan vs {ML}-pt map, k=5, 1 GPU (BASELINE configs[3] sizes un-sharded)", "passes_per_update": int(pl_),
                         "us_per_update": us, "iters_per_s": pl_ / (us * 1e-6), "alg_bytes_per_point_pass": b_alg(ML),
                         "whole_update_frac": b_alg(ML) * NL * pl_ / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "one_launch_per_pass": bool(c3.last_update_fused())}
            del scl
        except Exception as e:  # noqa: BLE001
            large = {"error": str(e)}
    # ---- BASELINE configs[1] and configs[2], driver-timed, each in a context of its own with the parity counts
    # of THAT leg against the oracle: cfg1 = a 16-ring VLP-16 scan (~30 k points, ray-cast) against a 500 k-point map, ONE
    # measurement pass (MAX_NUM_ITERS = 0: "single iteration"); cfg2 = a 64-ring scan (~120 k points) against a 2 M-point map,
    # 4 passes.  Same synchronised step as `value`; fraction = algorithmic bytes of all passes / wall time of a step / 8 TB/s.
    cfg_legs = {}
    if world == 1 and not args.no_cfgs and not args.resident_only:
        for name, (mm, rings, n_az, fov, iters, what) in {
                "cfg1": (500_000, 16, 1875, (-15.0, 15.0), 0, "BASELINE configs[1]: VLP-16 ring scan vs 500k-pt map, k=5, single iteration"),
                "cfg2": (2_000_000, 64, 2048, (-25.0, 15.0), 3, "BASELINE configs[2]: 64-line ring scan vs 2M-pt map, 4 IKFoM passes")}.items():
            try:
                scc = synth.make_ring_scene(mm, rings, n_az, fov_deg=fov)
                nn = len(scc["scan_xyz"])
                with capi.Context(capi.default_params(MAX_NUM_ITERS=iters), device=local_rank) as c5:
                    c5.map_build(scc["map_xyz"])
                    c5.scan_set(scc["scan_xyz"])
                    xc = np.ascontiguousarray(scc["x_init"], np.float64); Pc = np.ascontiguousarray(scc["P0"], np.float64)
                    xcp, Pcp = xc.ctypes.data_as(C.c_void_p), Pc.ctypes.data_as(C.c_void_p)
                    for _ in range(10):
                        c5.filter_set(xc, Pc); c5.correct(want_passes=False); c5.filter_get()
                    rates = []
                    for _ in range(3):
                        c5.synchronize(); t0 = time.perf_counter()
                        for _ in range(args.steps):
                            if c5.lib.lv_filter_set(c5.h, xcp, Pcp) or c5.lib.lv_correct(c5.h, None) or c5.lib.lv_filter_get(c5.h, xgp, Pgp):
                                raise RuntimeError(c5.lib.lv_last_error().decode())
                        c5.synchronize()
                        rates.append((time.perf_counter() - t0) / args.steps)
                    pc_ = c5.last_passes()
                    us = sorted(rates)[1] * 1e6
                    leg = {"workload": f"{what} ({nn}-pt scan)", "scan_points": int(nn), "map_points": int(mm), "passes_per_update": int(pc_),
                           "us_per_update": us, "iters_per_s": pc_ / (us * 1e-6), "alg_bytes_per_point_pass": b_alg(mm),
                           "frac": b_alg(mm) * nn * pc_ / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "one_launch_per_pass": bool(c5.last_update_fused())}
                    if not args.no_parity:
                        # the leg's own gate: one capturing pass + the timed build's update, against the oracle on the same inputs
                        sys.path.insert(0, os.path.join(ROOT, "oracle"))
                        import lvoracle as lo
                        g0 = c5.iterate(scc["x_init"])
                        gi, gd = c5.fetch_knn()
                        xg_, Pg_, pg_, trg_, sumsg_ = c5.update(scc["x_init"], scc["P0"])
                        tree = lo.KdTree(scc["map_xyz"])
                        o = lo.iterate(scc["x_init"], scc["map_xyz"], scc["scan_xyz"], tree=tree, nthreads=16)
                        xo_, Po_, po_, tro_, so_ = lo.update(scc["x_init"], scc["P0"], scc["map_xyz"], scc["scan_xyz"],
                                                             params=lo.default_params(max_num_iters=iters), tree=tree, nthreads=16)
                        par = {"knn_index_mismatches": int((gi != o["knn_idx"]).any(axis=1).sum()),
                               "knn_distance_bit_mismatches": int((gd.view(np.uint32) != o["knn_d2"].view(np.uint32)).any(axis=1).sum()),
                               "n_valid_gpu_oracle": [int(g0["n_valid"]), int(o["n_valid"])],
                               "HTH_rel_diff": float(np.abs(g0["HTH"] - o["HTH"]).max() / max(np.abs(o["HTH"]).max(), 1e-300))}
                        if xo_ is not None:
                            par.update({"passes_gpu_oracle": [int(pg_), int(po_)], "state_max_abs_diff": float(np.abs(xg_ - xo_).max()),
                                        "cov_max_abs_diff": float(np.abs(Pg_ - Po_).max())})
                        par["ok"] = bool(par["knn_index_mismatches"] == 0 and par["knn_distance_bit_mismatches"] == 0
                                         and par["n_valid_gpu_oracle"][0] == par["n_valid_gpu_oracle"][1] and par["HTH_rel_diff"] < 1e-9
                                         and (xo_ is None or (par["passes_gpu_oracle"][0] == par["passes_gpu_oracle"][1] and par["state_max_abs_diff"] < 1e-7)))
                        leg["parity"] = par
                    cfg_legs[name] = leg
                del scc
            except Exception as e:  # noqa: BLE001
                cfg_legs[name] = {"error": str(e)}
    # ---- the shipped non-default configuration (config/xaloc.yaml:13 estimate_extrinsics: true; 12 live Jacobian columns, 92 sums,
    # 12 x 12 gain blocks) at the headline sizes, same synchronised step, a context of its own: it/s + roofline fraction
    ext_rec = None
    if world == 1 and not args.no_ext and not args.resident_only and not args.extrinsics:
        try:
            sce = synth.make_scene(M_POINTS, N_POINTS, extrinsics="xaloc")
            with capi.Context(capi.default_params(estimate_extrinsics=1), device=local_rank) as c4:
                c4.map_build(sce["map_xyz"])
                c4.scan_set(sce["scan_xyz"])
                xe = np.ascontiguousarray(sce["x_init"], np.float64); Pe = np.ascontiguousarray(sce["P0"], np.float64)
                xep, Pep = xe.ctypes.data_as(C.c_void_p), Pe.ctypes.data_as(C.c_void_p)
                for _ in range(10):
                    c4.filter_set(xe, Pe); c4.correct(want_passes=False); c4.filter_get()
                rates = []
                for _ in range(3):
                    c4.synchronize(); t0 = time.perf_counter()
                    for _ in range(args.steps):
                        if c4.lib.lv_filter_set(c4.h, xep, Pep) or c4.lib.lv_correct(c4.h, None) or c4.lib.lv_filter_get(c4.h, xgp, Pgp):
                            raise RuntimeError(c4.lib.lv_last_error().decode())
                    c4.synchronize()
                    rates.append((time.perf_counter() - t0) / args.steps)
                pe = c4.last_passes()
                c4.set_profiling(True)
                k_ms, cnt = 0.0, 0
                for _ in range(min(args.steps, 50)):
                    _, _, p_, _, _ = c4.update(xe, Pe, want_trace=False)
                    k_ms += c4.timing()["last_reduce_ms"] * p_
                    cnt += p_
                c4.set_profiling(False)
                us = sorted(rates)[1] * 1e6
                k_us = k_ms / max(cnt, 1) * 1e3
                ext_rec = {"workload": "estimate_extrinsics = true (config/xaloc.yaml), 65536-pt scan vs 1048576-pt map, xaloc extrinsics",
                           "passes_per_update": int(pe), "us_per_update": us, "iters_per_s": pe / (us * 1e-6), "avg_kernel_us": k_us,
                           "frac": (b_alg(M_POINTS) * N_POINTS / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if k_us > 0 else None,
                           "one_launch_per_pass": bool(c4.last_update_fused())}
            del sce
        except Exception as e:  # noqa: BLE001
            ext_rec = {"error": str(e)}
    # ---- parity gate, GPU side (every rank: with a communicator the update is a collective): one capturing pass over this
    # rank's shard at the initial state + the timed build once more with its per-pass log; rank 0 then checks against the oracle
    gate = None
    if not args.no_parity:
        try:
            g0 = ctx.iterate(sc["x_init"])
            g_idx, g_d2 = ctx.fetch_knn()
            g_valid, _, g_abcd, g_dist = ctx.fetch_matches()
            if world == 1 or lib_comm:
                xg, Pg, pg, trg, sumsg = ctx.update(sc["x_init"], sc["P0"])
            else:
                xg, Pg, pg = upd.update(sc["x_init"], sc["P0"])
                trg, sumsg = None, None
            gate = dict(g0=g0, idx=g_idx, d2=g_d2, valid=g_valid, abcd=g_abcd, dist=g_dist, x=xg, P=Pg, passes=pg, tr=trg, sums=sumsg,
                        x_res=x_res, P_res=P_res)
        except Exception as e:  # noqa: BLE001
            gate = {"error": str(e)}
    cycle = None
    if world == 1 and not args.no_cycle:
        try:
            cycle = cycle_64k(ctx, sc, capi)     # (last: it inserts into the map)
        except Exception as e:  # noqa: BLE001
            cycle = {"error": str(e)}

    rc = 0
    if rank == 0:
        value = total_passes / dt
        # which kernel ran the passes: one launch per pass (pass_kernel: the solve of the previous pass in every workgroup +
        # search + plane fits) or the three-kernel pass (search / fit / solve)
        fused = fused_main
        kname = "pass" if fused else "search"
        avg_kernel_s = (kern_ms / max(kern_cnt, 1)) * 1e-3
        alg_bytes = b_alg(M_POINTS) * n_local
        achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        traffic, traffic_file = pmc_traffic_bytes(kname) if world == 1 else (None, None)
        rp_us, rp_file = rocprof_kernel_avg_us("lv::pass_kernel<true, false" if prm.estimate_extrinsics else "lv::pass_kernel<false, false")   # (name prefixes: the instantiation has a third argument since round 4)
        if forms and same_dev:
            forms["peer_mapped_one_launch"].update(iters_per_s=value, ms_per_step=dt / args.steps * 1e3)
        elif forms:
            forms["allgather_one_launch" if fused else "allreduce_three_kernel"].update(iters_per_s=value, ms_per_step=dt / args.steps * 1e3)
        out = {
            "metric": "KF-update iters/sec, 64k-pt scan vs 1M-pt map",
            "value": value,
            "unit": "KF-update iters/s",
            "n_gpus": 1 if same_dev else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "step": ("lv_filter_set + lv_correct + lv_filter_get on the device-resident filter: the posterior is read back after every "
                     "update, one host wait per step (the reference's call pattern, src/main.cpp:84-86)" if resident
                     else "lv_update by value (one host round trip per step)"),
            "value_definition": "round 5: synchronised step (posterior read every step), like rounds 1-3 and unlike round 4, whose `value` "
                                "was the pipelined rate now reported as value_pipelined",
            "value_pipelined": None if not pipelined else pipelined["value"],
            "value_by_value": None if not by_value_sync else by_value_sync["value"],
            "value_pipelined_detail": pipelined,
            "value_by_value_sync": by_value_sync,
            # the R timed regions of K steps each: value / ms_per_step are the median region's
            "regions": len(region_dt),
            "value_min": min(p / d for p, d in zip(region_passes, region_dt)),
            "value_max": max(p / d for p, d in zip(region_passes, region_dt)),
            "value_per_region": [round(p / d, 1) for p, d in zip(region_passes, region_dt)],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (kNN/plane fit) + f64 (Jacobian rows, H^T H, 23-dof solve)",
            "data": "synthetic",
            "config": {
                "workload": "iterated KF update: 65536-pt scan vs 1048576-pt map, k=5, MAX_NUM_ITERS=3 (4 passes/update)",
                "passes_per_update": total_passes / args.steps,
                "points_per_gpu": n_local,
                "parallelism": f"scan points sharded x{world}, map replicated, one collective per pass (see collective)" if world > 1 else "1 GPU",
                "collective": collective,
                "estimate_extrinsics": bool(prm.estimate_extrinsics),
                "lanes_per_query": prm.lanes_per_query,
                "voxel_size": prm.voxel_size,
            },
            # SURVEY 8(d) metric 2, "kNN Mpts/s": by the whole searching launch (solve of the previous pass + search + plane fits) and
            # by the search phase alone (in-kernel stamps of a converged launch, instrumented context)
            "knn_mpts_per_s": n_local * world / avg_kernel_s / 1e6 if avg_kernel_s > 0 else None,
            "knn_mpts_per_s_search_phase": (n_local / (phases["per_launch_us"]["search"][-1] * 1e-6) / 1e6) if phases else None,
            "roofline": {
                "bound": "hbm",
                "kernel": "lv::pass_kernel (one launch per pass: solve of the previous pass + search + plane fits)" if fused else "lv::search_kernel",
                "launches_per_update": (int(round(total_passes / args.steps)) + 1) if fused else 3 * int(round(total_passes / args.steps)),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the same fraction with the kernel's average duration from the committed rocprofv3 --kernel-trace --stats
                # summary of this command (HIP events around a kernel add ~2 us to what they bracket)
                "frac_rocprof": (alg_bytes / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (rp_us and world == 1 and fused) else None,
                "frac_rocprof_source": None if not (rp_us and world == 1 and fused) else
                f"from_committed_profile: profiles/{rp_file} (avg {rp_us:.2f} us per launch; not this run)",
                # HBM bytes per launch from the PMC counters: NOT measured in this run (rocprofv3 wraps a process; it cannot be
                # started from inside one) but read from the committed summary of the same command (scripts/gpu_profile.sh)
                "traffic": traffic,
                "traffic_source": None if traffic is None else f"from_committed_profile: profiles/{traffic_file} (PMC pass of this command, not this run)",
                # the same kernel priced with the bytes it really moves instead of the algorithmic ones
                "frac_measured_bytes": (traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_kernel_s > 0) else None,
                "measured_traffic_gbs": (traffic / avg_kernel_s / 1e9) if (traffic and avg_kernel_s > 0) else None,
                "alg_bytes_per_launch": alg_bytes,
                "alg_bytes_per_point_pass": b_alg(M_POINTS),
                "avg_kernel_us": avg_kernel_s * 1e6,
                # the same HIP-event durations by launch index within an update: the first launch (pose off by the perturbation: more
                # open points, unbalanced tiles) and the converged ones are two regimes — rocprofv3's average mixes them
                "kernel_us_by_launch": [round(float(u / max(n_, 1)), 2) for u, n_ in zip(per_launch_us[:4], per_launch_n[:4])],
                "kernel_us_first": float(per_launch_us[0] / max(per_launch_n[0], 1)),
                "kernel_us_converged": float(per_launch_us[1:4].sum() / max(per_launch_n[1:4].sum(), 1)),
                "frac_converged": (alg_bytes / (per_launch_us[1:4].sum() / max(per_launch_n[1:4].sum(), 1) * 1e-6) / 1e9 / HBM_PEAK_GBS)
                if per_launch_n[1:4].sum() > 0 and per_launch_us[1:4].sum() > 0 else None,
                "avg_fit_plus_solve_us": solve_ms / max(kern_cnt, 1) * 1e3,
                "avg_solve_us": solve_ms / max(kern_cnt, 1) * 1e3,
                "last_update_match_us_per_pass": [round(v * 1e3, 1) for v in ctx.timing()["pass_match_ms"][:4]],
                # SURVEY 8(d) metric 3: algorithmic bytes of ALL passes of a step over the step's wall time
                "whole_update_frac": alg_bytes * (total_passes / args.steps) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                "cold": cold,
                # inside pass_kernel (in-kernel wall-clock stamps, separate instrumented context): the search phase alone —
                # what round 1's search_kernel figure measured, minus launch ramp and the record stores that no longer exist
                "pass_kernel_phases": phases,
                "search_phase": None if not phases else {
                    "us_converged": phases["per_launch_us"]["search"][-1],
                    "alg_gbs": alg_bytes / (phases["per_launch_us"]["search"][-1] * 1e-6) / 1e9,
                    "alg_frac": alg_bytes / (phases["per_launch_us"]["search"][-1] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "traffic_gbs": (traffic / (phases["per_launch_us"]["search"][-1] * 1e-6) / 1e9) if traffic else None,
                },
            },
            "fallback": ctx.timing()["fallback_queries"],
            "state_check": {"pos_err_m_from_ground_truth": float(np.linalg.norm(x[:3] - sc["x_true"][:3]))},
        }
        if world > 1 or forms:
            out["multi_gpu"] = {"measured_on": f"{world} ranks" + (" on ONE GPU (--same-device)" if same_dev else ""),
                                "ranks": rank_report,
                                "note_scaling": "no scaling curve has been measured by the builder (one GPU per lease in rounds 1-6): the driver's "
                                                "N = 1, 2, 4, 8 runs of this file are the first executions across xGMI",
                                "collective_us_per_pass": [round(float(v), 2) for v in coll_us[:4]], "forms": forms,
                                "predicted": predicted_scaling(world, N_POINTS)}
            if same_dev:
                out["multi_gpu"]["same_device"] = True
                out["multi_gpu"]["note"] = ("FUNCTIONAL leg: the ranks share one GPU (and its L2) and time-slice it — value is NOT a "
                                            "scaling measurement; what it shows: self-launch, sharding, the peer-mapped exchange kernel's "
                                            "time per pass, bitwise-equal ranks")
        if cycle is not None:
            out["cycle_ms_64k"] = cycle
        if large is not None:
            out["large_n"] = large
        for kname_, leg_ in cfg_legs.items():
            out[kname_] = leg_
            if isinstance(leg_, dict) and isinstance(leg_.get("parity"), dict) and not leg_["parity"].get("ok", True):
                rc = 3   # (a leg whose results differ from the oracle's fails the run like the headline's gate)
        if ext_rec is not None:
            out["ext"] = ext_rec
        if gate is not None:
            if "error" in gate:
                out["parity"] = {"ok": False, "error": gate["error"]}
            else:
                out["parity"] = parity_check(gate, sc, upd_scan_local(sc, rank, world), prm)
            if not out["parity"]["ok"]:
                rc = 3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, int(total_passes / args.steps))
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank_report is not None and rank == 0 and (not rank_report["ranks_bitwise_equal"] or rank_report["points_total"] != N_POINTS):
        print("bench.py: RANKS DISAGREE (see multi_gpu.ranks in the JSON line)", file=sys.stderr)
        rc = rc or 4
    if rc:
        print("bench.py: PARITY GATE FAILED (see \"parity\" in the JSON line)", file=sys.stderr)
        raise SystemExit(rc)


if __name__ == "__main__":
    main()
