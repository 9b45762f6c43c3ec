"""CPU oracle of the Gaussian-rasterizer hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package; the product (``structured-gaussian-splatting_amd/``) never does.
Parity status: see the header of ``oracle/gsr_oracle.c`` (sub-steps with an in-tree twin are
pinned by ``tests/golden``; the rasterizer proper is "parity unpinned").
"""
from .gsr_oracle import OracleFrame, OracleRawFrame, build, rasterize, rasterize_raw, dist2_knn3, lib_path  # noqa: F401
