import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch ships its own ROCm runtime libraries; when a process uses both torch.cuda and liblimovelo_hip.so,
    # torch's runtime has to be initialised FIRST (the later-loaded copy reuses the already-loaded libamdhip64;
    # the other order leaves torch without devices).  bench.py imports torch first for the same reason.
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def lv():
    import lvamd

    return lvamd.load()


@pytest.fixture(scope="session")
def oracle():
    import lvoracle

    lvoracle.build()
    return lvoracle


@pytest.fixture(scope="session")
def scene_small(lv):
    """configs[0]: 2k-pt scan vs 50k-pt map."""
    from limo_velo_amd import synth

    return synth.make_scene(50_000, 2_000)
