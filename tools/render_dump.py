"""python tools/render_dump.py OUT.npy [gl_iter] [GiB]: renders the 720x720 Cornell box (seed 1234) and saves the image (needs a GPU)."""
import sys
from pathlib import Path

import numpy as np

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
import cudaraytracer_amd as crt  # noqa: E402

gl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
st = crt.make_settings(gl_iter=gl)
ctx = crt.Context(0)
if len(sys.argv) > 3:
    ctx.set_workspace_limit(int(float(sys.argv[3]) * (1 << 30)))
ctx.upload(crt.cornell_box(50))
ctx.rng_init(st.nx, st.ny, 1234)
fb = ctx.render_to_host(st)
np.save(sys.argv[1], fb)
print(sys.argv[1], fb.shape, float(fb[..., :3].mean()))
ctx.close()
