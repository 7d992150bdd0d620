"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly what
include/limovelo_hip.h declares; struct layouts seen by the ctypes binding match the C side; and the
product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "limovelo_hip.h")


@pytest.fixture(scope="module")
def capi(lv):
    from limo_velo_amd import capi as c

    if not os.path.exists(c.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return c


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lv_[A-Za-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree(capi):
    declared = _declared_functions()
    assert declared == sorted(capi.ABI_SYMBOLS), set(declared) ^ set(capi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(capi):
    lib = capi.load_library()
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} is declared in limovelo_hip.h but not exported"
    assert b"gfx950" in lib.lv_version()


def test_struct_layouts_match_c(capi, tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "limovelo_hip.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(lv_params), offsetof(lv_params, LIMITS),"
        " offsetof(lv_params, voxel_size), offsetof(lv_params, lanes_per_query), sizeof(lv_state), sizeof(lv_sums),"
        " offsetof(lv_sums, n_valid), sizeof(lv_timing), sizeof(lv_cloud_format), offsetof(lv_cloud_format, relative_time),"
        " sizeof(lv_ingest_params), offsetof(lv_ingest_params, full_rotation_time), offsetof(lv_ingest_params, min_dist),"
        " sizeof(lv_motion_state), offsetof(lv_params, degeneracy_mode), offsetof(lv_params, print_degeneracy_values),"
        " offsetof(lv_timing, mailbox_resyncs), sizeof(lv_map_stats), offsetof(lv_map_stats, bytes));return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(capi.Params), capi.Params.LIMITS.offset, capi.Params.voxel_size.offset,
            capi.Params.lanes_per_query.offset, 26 * 8, C.sizeof(capi.Sums), capi.Sums.n_valid.offset,
            C.sizeof(capi.Timing), C.sizeof(capi.CloudFormat), capi.CloudFormat.relative_time.offset,
            C.sizeof(capi.IngestParams), capi.IngestParams.full_rotation_time.offset, capi.IngestParams.min_dist.offset, 184,
            capi.Params.degeneracy_mode.offset, capi.Params.print_degeneracy_values.offset,
            capi.Timing.mailbox_resyncs.offset, C.sizeof(capi.MapStats), capi.MapStats.bytes.offset]
    assert got == want


def test_default_params_mirror_reference_yaml(capi):
    p = capi.default_params()  # config/params.yaml:32,46-53; src/main.cpp:145
    assert (p.MAX_NUM_ITERS, p.NUM_MATCH_POINTS, p.MAX_DIST_PLANE, p.estimate_extrinsics) == (3, 5, 2.0, 0)
    assert abs(p.PLANES_THRESHOLD - 0.05) < 1e-9 and p.LiDAR_noise == 0.001
    assert list(p.LIMITS) == [0.001] * 23


def test_no_gpu_means_loud_failure_not_fallback(capi):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.LvError) as e:
        capi.Context()
    assert "HIP device" in str(e.value) or "-3" in str(e.value)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under limo-velo_amd/ (or the ABI header) may import,
    link or call it."""
    pkg = os.path.join(ROOT, "limo-velo_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"lvoracle|lv_oracle|liblvoracle|lvo_", text):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_cpp_shim_builds_with_reference_method_names(capi):
    """g++ compiles and links the Mapper / Localizator shim (no GPU needed) and it exports the reference's
    method names (Mapper.hpp:9-38, Localizator.hpp:12-45)."""
    host = os.path.join(ROOT, "limo-velo_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    syms = subprocess.check_output(["nm", "-DC", os.path.join(ROOT, "limo-velo_amd", "liblimovelo_shim.so")], text=True)
    for name in ("Mapper::add(", "Mapper::match(", "Mapper::exists()", "Mapper::size()", "Mapper::hasToMap(",
                 "Localizator::correct(", "Localizator::calculate_H(", "Localizator::latest_state()",
                 "Localizator::propagate_to(", "Localizator::propagate(", "Compensator::compensate(", "State::operator+=("):
        assert name in syms, name


def test_pass_kernel_geometry_policy(capi):
    """pass_grid_size (host logic of the one-launch-per-pass kernel): every 32-point tile of the scan has exactly one owner
    slot, never more searching workgroups than CUs, one search step while a step per workgroup covers the scan, a dedicated
    bookkeeping workgroup exactly when a CU is left over, and the sizes at which lv_update switches routes."""
    for cus in (8, 64, 256, 304):
        for n in list(range(1, 300, 7)) + [1000, 4096, 32 * 4 * cus - 1, 32 * 4 * cus, 32 * 4 * cus + 1, 32 * 8 * (cus - 1), 32 * 8 * (cus - 1) + 1,
                                            32 * 8 * cus, 32 * 8 * cus + 1, 32 * 16 * cus, 32 * 16 * cus + 1, 1_000_000]:
            g, steps, rounds, ded = capi.pass_geometry(n, cus)
            tiles = (n + 31) // 32
            assert 1 <= g <= cus and steps in (1, 2) and rounds >= 1 and ded in (0, 1)
            assert g * 4 * steps * rounds >= tiles                       # every tile has a slot ...
            assert g * 4 * steps * (rounds - 1) < tiles                   # ... and no round is empty
            if rounds == 1:
                assert (g - 1) * 4 * steps < tiles                        # no workgroup without a tile
            assert (steps == 1) == (tiles <= 4 * cus)
            assert ded == (1 if g < cus else 0)
            assert g + ded <= cus                                        # everything resident at once: one workgroup per CU
    assert capi.pass_geometry(65_536, 256) == (256, 2, 1, 0)             # the headline: every CU searches, the books follow the fits
    assert capi.pass_geometry(65_280, 256) == (255, 2, 1, 1)
    assert capi.pass_geometry(32_768, 256) == (256, 1, 1, 0)
    assert capi.pass_geometry(8_192, 256) == (64, 1, 1, 1)
    assert capi.pass_geometry(131_072, 256)[2] == 2 and capi.pass_geometry(131_073, 256)[2] == 3
    # lv_update takes one launch per pass up to 16 rounds per workgroup (3 with estimate_extrinsics): the sizes of that rule
    assert capi.pass_geometry(196_608, 256)[2] == 3 and capi.pass_geometry(196_609, 256)[2] == 4
    assert capi.pass_geometry(262_144, 256)[2] == 4                      # configs[3]'s scan
    assert capi.pass_geometry(1_048_576, 256)[2] == 16 and capi.pass_geometry(1_048_577, 256)[2] == 17
