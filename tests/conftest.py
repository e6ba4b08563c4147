"""Shared pytest fixtures.  `-m "not gpu"` runs everywhere; `-m gpu` needs one MI355X."""
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU-side cases (still part of the default run)")


@pytest.fixture(scope="session")
def oracle():
    """The plain-C restatement (oracle/aqe_oracle.c) — the checker, never the product."""
    from oracle.pyoracle import Oracle, build
    build(ref=False)
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    """Outputs of the reference's own C++ recorded by oracle/make_golden.py."""
    return json.loads((ROOT / "tests" / "golden" / "ref_golden.json").read_text())


_TABLES = {}


@pytest.fixture(scope="session")
def table(oracle):
    """table(N, seed=42) -> synthetic rows (numpy structured array), cached per session."""
    def get(n, seed=42):
        key = (n, seed)
        if key not in _TABLES:
            if len(_TABLES) > 6:
                _TABLES.clear()
            _TABLES[key] = oracle.synth(n, seed)
        return _TABLES[key]
    return get
