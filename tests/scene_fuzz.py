"""Seeded random scenes for parity tests: spheres of every material that may overlap each other and the
walls, tilted parallelograms, trapezoids, general and non-planar quadrilaterals (small ones too), one or two
area lights of arbitrary orientation (parallelogram or not, one- or two-sided), sometimes a point light,
sometimes a missing wall.

The same POD records build the product's scene (mod = cudaraytracer_amd) and the checker's (mod = the
oracle wrapper), so both sides see identical floats."""
import numpy as np

import cudaraytracer_amd as crt
from cudaraytracer_amd import abi


def _set3(dst, v):
    dst[0], dst[1], dst[2] = float(v[0]), float(v[1]), float(v[2])


def _unit(v):
    return v / np.linalg.norm(v)


def _random_frame(rng, lo, hi):
    """Two edge vectors of a random parallelogram with side lengths in [lo, hi]."""
    while True:
        a, b = rng.normal(size=3), rng.normal(size=3)
        n = np.cross(a, b)
        if np.linalg.norm(n) > 0.3 * np.linalg.norm(a) * np.linalg.norm(b):
            return _unit(a) * rng.uniform(lo, hi), _unit(b) * rng.uniform(lo, hi)


def random_scene(mod, seed, shadow_samples=6, wild=False):
    """wild: also a random scale of the whole scene (the reference's 0.01 ray offsets do not scale with it), a
    camera that is moved, turned or orthographic, lights anywhere in the box, and mostly glass."""
    rng = np.random.default_rng(seed)
    wrng = np.random.default_rng(seed + 7_000_000)
    scale = 1.0
    glassy = False
    if wild:
        scale = float(wrng.choice([0.05, 0.3, 1.0, 1.0, 4.0, 50.0]))
        glassy = wrng.random() < 0.4
    base = crt.cornell_box(shadow_samples)
    clone = lambda g: abi.crt_geometry.from_buffer_copy(bytes(g))
    mirror, glass, diffuse = base.geometry[0].mat, base.geometry[1].mat, [g.mat for g in base.geometry[2:7]]

    def material():
        kind = rng.integers(0, 3)
        if glassy and wrng.random() < 0.7:
            kind = 2
        if kind == 0:
            return diffuse[rng.integers(0, len(diffuse))]
        if kind == 1:
            return mirror
        m = abi.crt_material.from_buffer_copy(bytes(glass))
        m.refractIndex = float(rng.uniform(1.05, 1.9))
        return m

    geo = []
    for k in range(2, 7):  # the five walls; now and then one is missing and rays leave the box
        if rng.random() > 0.12:
            geo.append(clone(base.geometry[k]))
    if rng.random() < 0.5:
        geo.append(clone(base.geometry[7]))  # the emissive panel under the stock light
    for _ in range(rng.integers(0, 5)):
        s = clone(base.geometry[0])
        _set3(s.center, (rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(-1.8, -0.4)))
        s.radius = float(rng.uniform(0.05, 0.45))
        s.mat = material()
        geo.append(s)
    for _ in range(rng.integers(0, 4)):
        q = clone(base.geometry[2])
        e1, e2 = _random_frame(rng, 0.05 if rng.random() < 0.3 else 0.2, 0.9)
        p0 = np.array((rng.uniform(-0.8, 0.4), rng.uniform(-0.8, 0.4), rng.uniform(-1.8, -0.6)))
        n = _unit(np.cross(e1, e2))
        shape = rng.random()
        a, b, lift = 1.0, 1.0, 0.0
        if shape > 0.9:      # corner 2 off the plane of the other three
            a, b, lift = rng.uniform(0.7, 1.3), rng.uniform(0.7, 1.3), rng.uniform(-0.1, 0.1)
        elif shape > 0.65:   # general quadrilateral
            a, b = rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4)
        elif shape > 0.4:    # trapezoid: two sides stay parallel
            a = rng.uniform(0.5, 1.5)
        corners = [p0, p0 + e1, p0 + a * e1 + b * e2 + lift * n, p0 + e2]
        for k in range(4):
            _set3(q.v[k].pos, corners[k])
            _set3(q.v[k].normal, n)
        q.mat = material()
        geo.append(q)

    lights = []
    for _ in range(rng.integers(1, 3)):
        l = abi.crt_light.from_buffer_copy(bytes(base.lights[0]))
        _set3(l.pos, (rng.uniform(-0.6, 0.6), rng.uniform(0.2, 0.95), rng.uniform(-1.6, -0.5)))
        if rng.random() < 0.5:  # horizontal, turned about the vertical axis
            phi = rng.uniform(0, np.pi)
            e1 = np.array((np.cos(phi), 0.0, np.sin(phi))) * rng.uniform(0.1, 0.6)
            e2 = np.array((np.sin(phi), 0.0, -np.cos(phi))) * rng.uniform(0.1, 0.6)
        else:
            e1, e2 = _random_frame(rng, 0.1, 0.6)
        a, b = (1.0, 1.0) if rng.random() < 0.7 else (rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4))
        p0 = -0.5 * (e1 + e2)
        pts = [p0, p0 + e1, p0 + a * e1 + b * e2, p0 + e2]
        for k in range(4):
            _set3(l.points[k], pts[k])
        if wild and wrng.random() < 0.5:
            _set3(l.pos, (wrng.uniform(-0.9, 0.9), wrng.uniform(-0.9, 0.9), wrng.uniform(-1.9, -0.3)))
        l.twoSided = int(rng.random() < 0.5)
        l.intensity = float(rng.uniform(2.0, 12.0))
        l.shadowSamples = shadow_samples
        lights.append(l)
    if rng.random() < 0.3:
        lights.append(abi.crt_light(abi.CRT_LIGHT_POINT, (float(rng.uniform(-0.5, 0.5)), 0.5, -1.0),
                                    (1.0, 0.9, 0.8, 1.0), 0.5))
    cam = abi.crt_camera.from_buffer_copy(bytes(base.camera))
    if wild:
        if wrng.random() < 0.5:
            _set3(cam.pos, (wrng.uniform(-0.7, 0.7), wrng.uniform(-0.7, 0.7), wrng.uniform(-1.2, 0.5)))
            _set3(cam.front, _unit(np.array((wrng.uniform(-0.6, 0.6), wrng.uniform(-0.6, 0.6), -1.0))))
        if wrng.random() < 0.2:
            cam.type = abi.CRT_CAM_ORTHOGRAPHIC
            cam.orthoScale = float(wrng.uniform(0.3, 1.2)) * scale
        for g in geo:
            for k in range(3):
                g.center[k] *= scale
            g.radius *= scale
            for v in g.v:
                for k in range(3):
                    v.pos[k] *= scale
        for l in lights:
            for k in range(3):
                l.pos[k] *= scale
                for pt in l.points:
                    pt[k] *= scale
        for k in range(3):
            cam.pos[k] *= scale
    # every fourth scene: its area lights take the horizon-clipping LTC branch (a generator of its own, so that the
    # scenes themselves stay what they were before this flag existed)
    crng = np.random.default_rng(seed * 7919 + 13)
    if crng.random() < 0.25:
        for l in lights:
            if l.kind == abi.CRT_LIGHT_AREA:
                l.horizonClip = 1
    return mod.Scene(cam, geo, lights)
