"""The HIP path against the renders the reference itself published for gl_iter >= 1.

The reference holds three such images with recorded settings (720x720, aa_iter=1, ref_iter=4, ind_rays=75,
shadowSamples=50; README.md:104-120): renders/render1.hdr and file75.hdr (gl_iter=2, two wall-clock seeds)
and renders/render2.hdr (gl_iter=3).  Their seeds are unknown and they were written by a --use_fast_math
build through stb's truncating 8-bit RGBE encoder, so the comparison is statistical: the frame rendered here
goes through the same encoder (crt_encode_hdr) and the same decoder as the reference's file, and then

  * the mean colour must agree within 5e-4 relative (north_star's bound for gl_iter >= 1 is 1e-3; the
    reference's own two gl_iter=2 seeds agree to 7.5e-5), and
  * the 16x16-block means (45x45x3 values) must stay within the band the reference's two gl_iter=2 renders
    show against each other (max 0.0058, mean 0.00086 -- tests/golden/reference_hdr_stats.json, "gl2_seed_band"),
    with the factor stated below for the maximum of 6075 noisy values.  Measured here over four seeds: mean
    colour within 1.5e-4 relative (four seeds of this renderer scatter by 1.6e-4 among themselves), block
    means 0.0057-0.0077 max and 0.00083-0.00087 mean.
  * Blocks on the two image diagonals are held to a wider bound.  Pixel samples there lie exactly on the
    projected edge between two walls (or a wall and the floor): the two planes tie, the last bit of two
    quotients decides, and an IEEE build rounds them differently from the reference's --use_fast_math build.
    At gl_iter=0 this is visible in isolation (tests/test_oracle_golden.py: 193 of the 201 pixels that are off
    by more than 0.05 lie on the diagonals, a fixed 0.0046 offset of those blocks for every seed); the bound
    adds that offset to the band.

This is what ties the global-illumination expansion (main.cu:172-192: hemisphere weight 2*r1*diffuse/ind_rays,
the 128-entry stack truncation, the draw order) to something the reference holds rather than to the CPU
restatement alone.
"""
import json
from pathlib import Path

import numpy as np
import pytest

import cudaraytracer_amd as crt
from hdr_codec import decode_hdr, rgbe_to_float

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
STATS = json.loads((GOLD / "reference_hdr_stats.json").read_text())
MEAN_REL_TOL = 5e-4  # north_star allows 1e-3; three times the scatter between seeds is enough
BLOCK_MAX_FACTOR = 1.5   # the band is itself the maximum of one draw of 6075 differences
BLOCK_MEAN_FACTOR = 1.25
DIAGONAL_TIE_OFFSET = 0.005  # deterministic block offset of the tied edge samples, measured at gl_iter=0 (0.0046)


def _render_as_the_reference_writes_it(gl, seed):
    st = crt.make_settings(gl_iter=gl)  # 720x720, aa_iter=1, ref_iter=4, ind_rays=75, shadowSamples=50
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(st.shadowSamples))
        ctx.rng_init(st.nx, st.ny, seed)
        fb = ctx.render_to_host(st)
    _, rgbe = decode_hdr(crt.encode_hdr(fb))  # stbi_write_hdr's quantisation (main.cu:856), then the test's reader
    return rgbe_to_float(rgbe)


def _blocks(img):
    h, w, _ = img.shape
    return img.reshape(h // 16, 16, w // 16, 16, 3).astype(np.float64).mean((1, 3))


@pytest.mark.parametrize("gl,files,seed", [(2, ("renders/render1.hdr", "file75.hdr"), 20250101),
                                           (3, ("renders/render2.hdr",), 20250102)])
def test_gl_render_matches_the_reference_published_image(gl, files, seed):
    img = _render_as_the_reference_writes_it(gl, seed)
    mean = img.reshape(-1, 3).astype(np.float64).mean(0)
    blocks = _blocks(img)
    band = STATS["gl2_seed_band"]
    for rel in files:
        ref = STATS[rel]
        ref_mean = np.array(ref["mean_rgb"])
        err = np.abs(mean / ref_mean - 1.0)
        print(f"gl_iter={gl} vs {rel}: mean {mean.tolist()} reference {ref_mean.tolist()} relative error {err.tolist()}")
        assert err.max() <= MEAN_REL_TOL, f"mean colour differs from {rel} by {err.max():.2e} relative"
        d = np.abs(blocks - np.array(ref["block16_mean"]))
        print(f"  block means: max abs {d.max():.5f} (band {band['max_abs']:.5f}), mean abs {d.mean():.5f} (band {band['mean_abs']:.5f})")
        i = np.arange(45)
        on_diag = np.zeros((45, 45), bool)
        on_diag[i, i] = on_diag[i, 44 - i] = True
        off, on = d[~on_diag].max(), d[on_diag].max()
        print(f"  off the diagonals: max {off:.5f}; on them: max {on:.5f}")
        assert off <= BLOCK_MAX_FACTOR * band["max_abs"], f"a 16x16 block differs from {rel} by {off:.4f}"
        assert on <= BLOCK_MAX_FACTOR * band["max_abs"] + DIAGONAL_TIE_OFFSET
        assert d.mean() <= BLOCK_MEAN_FACTOR * band["mean_abs"]
