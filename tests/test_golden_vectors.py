"""Function-level golden vectors (tests/golden/function_vectors.npz, written by tests/golden/make_vectors.py).

CPU: the oracle recomputes every vector and must reproduce the committed values, so that a change of an arithmetic
convention in oracle/pt_oracle.cpp cannot go unnoticed (the oracle and the device code have one author; without this they
could drift together).  GPU: the HIP path, through the C ABI, against the same committed values wherever the ABI exposes
the quantity (closest / any hits, Sobol' values, film positions, per-sample radiance)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import vectors  # noqa: E402


@pytest.fixture(scope="module")
def want():
    return dict(np.load(os.path.join(GOLDEN, "function_vectors.npz")))


def test_oracle_reproduces_the_committed_vectors(want):
    got = vectors.compute()
    assert sorted(got) == sorted(want)
    assert vectors.compare(got, want) == []


def test_vectors_hold_the_cases_they_are_meant_to(want):
    """The stored answers themselves: inclusive bounds of the triangle test, NaN slabs of the box test, tree size."""
    hit = want["tri_hit"] == 1
    assert 0.3 < hit.mean() < 0.95
    assert (want["tri_hit_far_end_at_t"][hit] == 1).all()                  # t <= max_dis is inclusive (src/shape.cpp:88)
    assert (want["tri_hit_far_end_before_t"][hit] == 0).all()
    b = want["tri_bary"][hit]
    assert (b[:, 1] >= 0).all() and (b[:, 2] >= 0).all() and (b[:, 1] + b[:, 2] <= 1).all()
    assert 0.05 < want["box_hit"].mean() < 0.95
    assert int(want["topo_veach_nodes"][0]) == 6183
    assert set(np.unique(want["li_dims"] - 2) % 1) == {0} and want["li_dims"].min() >= 2
    # every path draws 6 dimensions per bounce (+1 per roulette test): a path with k closest-hit rays that ended on a miss
    # has used 2 + 6 (k - 1) .. 2 + 7 (k - 1) dimensions
    assert want["occ_flag"].mean() > 0.05 and (want["occ_first"][want["occ_flag"] == 0] == -1).all()


@pytest.mark.gpu
def test_gpu_matches_the_committed_vectors(want, veach_gpu, veach_sd):
    from pt_hip import api
    for mode in (api.TRAVERSE_EXACT, api.TRAVERSE_FAST):
        h = veach_gpu.intersect(want["hit_rays"], mode)
        assert np.array_equal(h["prim"], want["hit_prim"])
        m = want["hit_prim"] >= 0
        assert np.array_equal(h["t"][m].view(np.uint32), want["hit_t"][m].view(np.uint32))
        assert np.array_equal(h["u"][m].view(np.uint32), want["hit_bary"][m, 1].view(np.uint32))
        assert np.array_equal(h["v"][m].view(np.uint32), want["hit_bary"][m, 2].view(np.uint32))
        occ, first = veach_gpu.occluded(want["occ_rays"], mode)
        assert np.array_equal(occ, want["occ_flag"])
        if mode == api.TRAVERSE_EXACT:
            assert np.array_equal(first, want["occ_first"])
    rows = want["sobol_rows"]
    rows = rows[rows[:, 0] == 11]                                          # this scene's sampler grid is 2048 wide
    dims = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 1023]
    pix = np.repeat(rows[:, 1:3].astype(np.int32), len(dims), 0)
    smp = np.repeat(rows[:, 3], len(dims))
    got = veach_gpu.sobol(pix, smp, np.tile(np.array(dims, np.uint32), len(rows)))
    assert np.array_equal(got.view(np.uint32).reshape(len(rows), -1), rows[:, 6:])
    pix = np.repeat(want["li_pixels"], 16, 0)
    smp = np.tile(np.arange(16, dtype=np.uint32), len(want["li_pixels"]))
    rgb, ps, _ = veach_gpu.li(pix, smp, traversal=api.TRAVERSE_FAST)
    assert np.array_equal(ps.view(np.uint32), want["li_pixel_sample"].view(np.uint32))
    err = np.abs(rgb - want["li_rgb"]).max(1) / (np.abs(want["li_rgb"]).max(1) + 1e-3)
    assert (err < 1e-3).mean() >= 0.998 and (err == 0).mean() >= 0.98   # identical in every bit: measured 99.3 - 100 %


@pytest.mark.gpu
def test_gpu_bare_functions_match_the_committed_vectors(want):
    """Triangle::intersect and AABB::intersect on their own (no tree around them), through the probe entry points: the
    10^4 adversarial pairs each of the committed vectors, bit for bit."""
    from pt_hip import api, synth
    soup = synth.random_soup(200, seed=5)                                  # the triangles the vectors were made on
    corners = soup.verts[soup.vi[want["tri_prim"]] - 1].reshape(-1, 9)
    hit, tuv = api.probe_triangle(corners, want["tri_rays"])
    assert np.array_equal(hit, want["tri_hit"])
    assert np.array_equal(tuv[:, 0].view(np.uint32), want["tri_t"].view(np.uint32))
    m = want["tri_hit"] == 1
    assert np.array_equal(tuv[m, 1].view(np.uint32), want["tri_bary"][m, 1].view(np.uint32))
    assert np.array_equal(tuv[m, 2].view(np.uint32), want["tri_bary"][m, 2].view(np.uint32))
    at = want["tri_rays"].copy()
    at[:, 7] = np.where(m, want["tri_t"], 1.0)
    before = at.copy()
    before[:, 7] = np.nextafter(at[:, 7], np.float32(0))
    assert np.array_equal(api.probe_triangle(corners, at)[0], want["tri_hit_far_end_at_t"])
    assert np.array_equal(api.probe_triangle(corners, before)[0], want["tri_hit_far_end_before_t"])
    assert np.array_equal(api.probe_box(want["box_boxes"], want["box_rays"]), want["box_hit"])

