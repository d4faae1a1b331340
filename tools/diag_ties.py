"""python tools/diag_ties.py [seed]: renders the 720x720 gl_iter 0 frame with the loaded library (CRT_HIP_LIB picks a
variant build), pushes it through the RGBE writer and compares it with the reference's renders/render0.hdr (the
copy under tests/golden): how many pixels differ by more than 0.05 and how many of those lie on the image's two
diagonals, where pixel-corner samples see an exact tie between two walls."""
import sys
from pathlib import Path

import numpy as np

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
import torch  # noqa: E402,F401
import cudaraytracer_amd as crt  # noqa: E402
from hdr_codec import decode_hdr, rgbe_to_float  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1234
ref = rgbe_to_float(decode_hdr((root / "tests/golden/render0.hdr").read_bytes())[1]).astype(np.float64)
st = crt.make_settings(gl_iter=0)
with crt.Context(0) as ctx:
    ctx.upload(crt.cornell_box(st.shadowSamples))
    ctx.rng_init(st.nx, st.ny, seed)
    fb = ctx.render_to_host(st)
mine = rgbe_to_float(decode_hdr(crt.encode_hdr(fb))[1]).astype(np.float64)
d = np.abs(mine - ref).max(-1)
big = np.argwhere(d > 0.05)
main = sum(1 for r, c in big if r == c)
anti = sum(1 for r, c in big if r + c == 720)
blocks = np.abs((mine - ref).reshape(45, 16, 45, 16, 3).mean((1, 3))).max(-1)
i = np.arange(45)
on = np.zeros((45, 45), bool)
on[i, i] = on[i, 44 - i] = True
print(f"seed {seed}: pixels off by > 0.05: {len(big)} (main diagonal {main}, anti-diagonal {anti}); "
      f"block means: off the diagonals max {blocks[~on].max():.5f}, on them max {blocks[on].max():.5f}")
