import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """the native pieces are built in-tree once per session (no-op when fresh)"""
    from cfs_spmv_amd import build
    lib = os.path.join(ROOT, "cfs_spmv_amd", "libcfs_hip.so")
    if not os.path.exists(lib):
        build.build_hip()
    if not os.path.exists(os.path.join(ROOT, "cfs_spmv_amd", "libcfs_synth.so")):
        build.build_synth()
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        build.build_oracle()
    if not os.path.exists(os.path.join(ROOT, "build", "test_spmv_mmf")):
        build.build_cxx()
    yield


def scaled_err(y, y_ref, absrow):
    """|y - y_ref| / max(|y_ref|, sum_j |a_ij||x_j|): the scale BASELINE.md's
    1e-12 (fp64) / 1e-5 (fp32) bounds are taken against."""
    import numpy as np
    den = np.maximum(np.abs(y_ref), absrow)
    den = np.where(den == 0, 1.0, den)
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - y_ref) / den)) if len(y) else 0.0
