import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ASSETS = os.path.join(ROOT, "assets")
MISS = 0xFFFFFFFF


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vra():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def ora():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def teapot64(vra):
    return vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 64)


@pytest.fixture(scope="session")
def teapot128(vra):
    return vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 128)


@pytest.fixture(scope="session")
def sphere64(vra):
    return vra.HostScene.sphere(64)


def assert_planes_equal(got, want, what=""):
    """Bit-exact comparison of the five output planes (depth compared as bit patterns)."""
    for plane in ("hit_index", "steps_lo", "steps_hi", "rgba8"):
        if plane in got and plane in want:
            g, w = np.asarray(got[plane]), np.asarray(want[plane])
            assert g.shape == w.shape, f"{what} {plane}: shape {g.shape} vs {w.shape}"
            bad = np.flatnonzero(g.reshape(-1) != w.reshape(-1))
            assert bad.size == 0, (f"{what} {plane}: {bad.size} of {g.size} differ, first at {bad[0]}: "
                                   f"got {g.reshape(-1)[bad[0]]} want {w.reshape(-1)[bad[0]]}")
    if "pt_end" in got and "pt_end" in want:
        g = np.asarray(got["pt_end"]).view(np.uint32).reshape(-1)
        w = np.asarray(want["pt_end"]).view(np.uint32).reshape(-1)
        bad = np.flatnonzero(g != w)
        assert bad.size == 0, f"{what} pt_end: {bad.size} of {g.size} differ (bit patterns), first at {bad[0]}"
    if "depth" in got and "depth" in want:
        g = np.asarray(got["depth"]).view(np.uint32).reshape(-1)
        w = np.asarray(want["depth"]).view(np.uint32).reshape(-1)
        bad = np.flatnonzero(g != w)
        assert bad.size == 0, f"{what} depth: {bad.size} of {g.size} differ (bit patterns), first at {bad[0]}"
