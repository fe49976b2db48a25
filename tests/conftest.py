"""pytest configuration: import paths, the `gpu` marker, shared engine fixtures."""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG_ROOT = ROOT / "chessvision-3lc_amd"
for p in (str(ROOT), str(PKG_ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_available() -> bool:
    import torch

    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently skip: only auto-skip when the user did
    # not ask for gpu tests explicitly.
    if "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="needs a GPU (select with -m gpu)")
    if not _gpu_available():
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """Several CPU tests call host-only entry points of the C-ABI library (contours, FEN decoding, the plain-C consumer): build it
    once per session when the tree has none yet (a fresh checkout; the GPU box receives the built file with the snapshot)."""
    lib = PKG_ROOT / "lib" / "libchessvision_hip.so"
    if not lib.exists():
        import __graft_entry__ as ge

        try:
            ge.build()
        except Exception as exc:                             # no hipcc on this box: the pure-Python tests still run; the tests that
            print(f"conftest: native build failed ({exc!r}); tests that load the library will fail on their own", file=sys.stderr)
    yield


@pytest.fixture(scope="session")
def engines():
    """{'f32' | 'f16' | 'f16x3': HipEngine} with small chunks (tests use small batches)."""
    from chessvision.hip_backend import HipEngine

    made = {p: HipEngine(precision=p, unet_chunk=2, resnet_chunk=128) for p in ("f32", "f16", "f16x3")}
    yield made
    for e in made.values():
        e.close()
