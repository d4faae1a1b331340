"""Checksums of a rendered frame (image and final generator states), whole and per band of rows, used to
compare full-size GPU renders with frozen renders of the CPU restatement (tests/golden/oracle_full_frames.json)."""
import hashlib

import numpy as np

BAND = 16


def digest(fb, state):
    """fb: (ny, nx, 4) float32; state: (ny*nx, words) uint32."""
    ny, nx = fb.shape[:2]
    st = np.ascontiguousarray(state).reshape(ny, nx, -1)
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    return {
        "image_sha256": hashlib.sha256(fb.tobytes()).hexdigest(),
        "state_sha256": hashlib.sha256(st.tobytes()).hexdigest(),
        "image_bands": [hashlib.sha256(fb[y:y + BAND].tobytes()).hexdigest()[:16] for y in range(0, ny, BAND)],
        "state_bands": [hashlib.sha256(st[y:y + BAND].tobytes()).hexdigest()[:16] for y in range(0, ny, BAND)],
    }
