"""The horizon-clipping branch of AreaLight::_evaluateLTC (AreaLight.hpp:104-214,298-334) in the CPU restatement.

No artefact of the reference exercises this branch (AreaLight::_clipless is true and has no setter), so the
sixteen-case clipping table is checked against an independent construction: Sutherland-Hodgman clipping of the
same quadrilateral against z >= 0 in float64, followed by the same edge integral summed round the clipped
polygon.  Parity of the HIP path with the restatement is tests/test_gpu_parity.py (horizon_clip scenes, fuzz)."""
import ctypes as C

import numpy as np
import pytest


def _edge_z(v1, v2):
    """AreaLight::_integrateEdgeVec(v1, v2).z() (AreaLight.hpp:72-98) in float64."""
    x = float(np.dot(v1, v2))
    y = abs(x)
    a = 0.8543985 + (0.4965155 + 0.0145206 * y) * y
    b = 3.4175940 + (4.1616724 + y) * y
    v = a / b
    ts = v if x > 0.0 else 0.5 / np.sqrt(max(1.0 - x * x, 1e-4)) - v
    return float(np.cross(v1, v2)[2] * ts)


def _reference_sum(quad, two_sided):
    poly = []
    for k in range(4):                       # Sutherland-Hodgman against the half space z > 0
        a, b = quad[k], quad[(k + 1) % 4]
        if a[2] > 0.0:
            poly.append(a)
        if (a[2] > 0.0) != (b[2] > 0.0):
            t = a[2] / (a[2] - b[2])
            poly.append(a + t * (b - a))
    if len(poly) < 3:
        return 0.0
    poly = [p / np.linalg.norm(p) for p in poly]
    s = sum(_edge_z(poly[k], poly[(k + 1) % len(poly)]) for k in range(len(poly)))
    return abs(s) if two_sided else max(0.0, s)


def _clipped(oracle, quad, two_sided):
    arr = ((C.c_float * 3) * 4)()
    for k in range(4):
        for i in range(3):
            arr[k][i] = float(quad[k][i])
    return oracle.lib().orc_ltc_clipped(C.byref(arr), int(two_sided))


def _config(quad):
    return sum(1 << k for k in range(4) if np.float32(quad[k][2]) > 0)


def test_every_clipping_case_matches_a_numeric_clipper(oracle):
    rng = np.random.default_rng(11)
    seen = {}
    total = close = 0
    for _ in range(4000):
        # a planar quadrilateral in general position around the origin's horizon
        centre = rng.uniform(-1.0, 1.0, 3) * (1.0, 1.0, 0.6)
        e1, e2 = rng.normal(size=3), rng.normal(size=3)
        e1 /= np.linalg.norm(e1)
        e2 -= e1 * np.dot(e1, e2)
        e2 /= np.linalg.norm(e2)
        a, b = rng.uniform(0.3, 1.5, 2)
        quad = np.array([centre - a * e1 - b * e2, centre + a * e1 - b * e2, centre + a * e1 + b * e2, centre - a * e1 + b * e2])
        quad = quad.astype(np.float32).astype(np.float64)
        if np.abs(quad[:, 2]).min() < 1e-3 or np.linalg.norm(quad, axis=1).min() < 0.05:
            continue                                     # keep clear of the sign test's and the normalisation's edges
        cfg = _config(quad)
        for two_sided in (False, True):
            got, want = _clipped(oracle, quad, two_sided), _reference_sum(quad, two_sided)
            # binary32 against float64: an edge between nearly opposite directions multiplies a small cross product
            # by a large 0.5 / sqrt(1 - x^2), which costs digits; a wrong table entry is off by 1e-2 and more
            assert got == pytest.approx(want, abs=5e-4), (cfg, two_sided, quad.tolist())
            total += 1
            close += abs(got - want) <= 2e-5
        seen[cfg] = seen.get(cfg, 0) + 1
    assert close >= 0.99 * total
    # every configuration a planar quadrilateral can produce (5 and 10 need opposite corners alone above the horizon)
    assert set(seen) == set(range(16)) - {5, 10}, sorted(seen)


def test_impossible_configurations_and_all_below_give_nothing(oracle):
    up, down = 0.5, -0.5
    for zs in ((up, down, up, down), (down, up, down, up), (down, down, down, down)):
        quad = np.array([(-1, -1, zs[0]), (1, -1, zs[1]), (1, 1, zs[2]), (-1, 1, zs[3])], dtype=np.float64)
        assert _clipped(oracle, quad, False) == 0.0 and _clipped(oracle, quad, True) == 0.0


def test_a_quad_above_the_horizon_is_the_plain_sum_of_its_edges(oracle):
    quad = np.array([(-0.5, -0.5, 1.0), (0.5, -0.5, 1.0), (0.5, 0.5, 1.0), (-0.5, 0.5, 1.0)], dtype=np.float64)
    unit = [q / np.linalg.norm(q) for q in quad]
    want = sum(_edge_z(unit[k], unit[(k + 1) % 4]) for k in range(4))
    assert want > 0.0
    assert _clipped(oracle, quad, False) == pytest.approx(want, abs=1e-6)
    assert _clipped(oracle, quad[::-1].copy(), False) == 0.0          # the other winding: negative, clamped
    assert _clipped(oracle, quad[::-1].copy(), True) == pytest.approx(want, abs=1e-6)
