"""Sobol' tables and sampler of the oracle (reference: src/sampler.cpp, src/sobolmatrices.cpp)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF_TABLES = "/root/reference/src/sobolmatrices.cpp"


@pytest.mark.skipif(not os.path.exists(REF_TABLES), reason="reference tree not present")
def test_generated_tables_equal_reference_tables(tmp_path):
    """The tables derived from the Joe-Kuo direction numbers are the reference's tables, entry for entry."""
    out = tmp_path / "t.inc"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_sobol_tables.py"), "--out", str(out), "--verify", REF_TABLES],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical" in r.stdout


def test_dimension0_is_van_der_corput():
    for i, want in [(0, 0.0), (1, 0.5), (2, 0.25), (3, 0.75), (4, 0.125), (5, 0.625)]:
        assert orc.sobol_sample(i, 0) == want


def test_dimension1_known_values():
    # second Sobol' dimension: generator columns 1000.., 1100.., 1010.. (Pascal triangle mod 2), natural index order
    assert [orc.sobol_sample(i, 1) for i in range(1, 5)] == [0.5, 0.75, 0.25, 0.625]


def test_sample_is_below_one():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        v = orc.sobol_sample(int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 1024)))
        assert 0.0 <= v <= float(np.float32(1) - np.float32(2 ** -24))


@pytest.mark.parametrize("m", [1, 3, 10, 11, 12])
def test_interval_to_index_lands_in_its_pixel(m):
    """sobolIntervalToIndex(m, frame, p) is the frame-th index whose first two dimensions fall into pixel p of a 2^m grid."""
    rng = np.random.default_rng(m)
    scale = 1 << m
    seen = {}
    for _ in range(300):
        px, py, frame = int(rng.integers(scale)), int(rng.integers(scale)), int(rng.integers(0, 5000))
        idx = orc.sobol_index(m, frame, px, py)
        assert idx >> (2 * m) == frame
        assert int(orc.sobol_sample(idx, 0) * scale) == px
        assert int(orc.sobol_sample(idx, 1) * scale) == py
        assert seen.setdefault(idx, (px, py, frame)) == (px, py, frame)   # injective


def test_sampler_agrees_with_scipy_generator():
    """An independent implementation of the same Joe-Kuo sequence: SciPy's generator (32-bit, unscrambled) walks the points
    in Gray-code order, so its n-th point is the natural-order point with index n ^ (n >> 1) that sobolSample
    (src/sampler.cpp:6-14) computes.  All 64 leading dimensions, 4096 points, equal as float32."""
    from scipy.stats import qmc
    d, n = 64, 4096
    pts = qmc.Sobol(d=d, scramble=False, bits=32).random(n).astype(np.float32)
    for i in range(0, n, 7):
        g = i ^ (i >> 1)
        got = np.array([orc.sobol_sample(g, k) for k in range(d)], np.float32)
        assert np.array_equal(got, pts[i]), i
