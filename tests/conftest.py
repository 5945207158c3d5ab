import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("SMAP_PLAN_CACHE", "0")      # tests build hundreds of schedules of throw-away weights: no blobs in $HOME (the cache has its own test)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A clean checkout has no libsmap_hip.so (built artefacts are git-ignored): build it once, in-tree.  An existing
    library is left alone (snapshots do not always keep mtimes; rebuilding is `python -m smap_amd.build`).  hipcc
    cross-compiles gfx950 without a GPU; when it is absent the tests that need the library fail loudly."""
    try:
        from smap_amd.build import OUT, build_lib
        if not os.path.exists(OUT):
            build_lib(force=False, verbose=False)
    except Exception as exc:                                   # noqa: BLE001
        print(f"[conftest] libsmap_hip.so not (re)built: {exc!r}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def smap_cfg():
    from helpers import make_cfg
    return make_cfg


@pytest.fixture(autouse=True)
def two_input_launches_in_test_schedules(monkeypatch):
    """The schedule builder keeps round 6's launches (conv_cat, conv_relusum, the tap-dot head) out of SMALL schedules (<= 2 frames of 512x832:
    batch 1 lives on split K and the deep-pipeline tile, smap_amd/engine.py).  The test schedules are small by construction (2 frames of
    64x96) and exist to exercise exactly those launches: forced on here; the tests of the batch-1 rules remove the switches again."""
    for k in ("SMAP_CAT", "SMAP_SKIPSUM", "SMAP_TAPHEAD"):
        if k not in os.environ:
            monkeypatch.setenv(k, "1")
