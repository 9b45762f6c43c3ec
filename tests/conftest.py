"""pytest configuration: markers + import paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (CPU only).
`-m gpu`: parity tests proper — the HIP path (through the C ABI) against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "structured-gaussian-splatting_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import oracle
    oracle.build()
    yield
