import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The libraries are build artefacts (git-ignored): compile them when they are missing or stale
    (hipcc cross-compiles gfx950 without a GPU; a fresh build takes about two minutes)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.build()
