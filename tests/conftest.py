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


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """Every failure is written down where it survives the run (tools/stress_suite.py loops the suite and keeps only
    what failed): node id, phase, the assertion's text, the TOPS_* environment."""
    outcome = yield
    rep = outcome.get_result()
    log = os.environ.get("TOPS_FAILURE_LOG")
    if log and rep.failed:
        import json
        import time
        with open(log, "a") as f:
            f.write(json.dumps({"nodeid": item.nodeid, "when": rep.when, "time": time.time(),
                                "longrepr": str(rep.longrepr)[-6000:],
                                "env": {k: v for k, v in os.environ.items() if k.startswith(("TOPS_", "FUZZ_"))}}) + "\n")
