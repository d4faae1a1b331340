"""Independent Radiance RGBE (.hdr) reader used by the tests (the product only writes this format)."""
import numpy as np


def decode_hdr(raw: bytes):
    """Returns (header text up to and including the resolution line, uint8 array [h, w, 4] of R,G,B,E)."""
    sep = raw.index(b"\n\n") + 2
    eol = raw.index(b"\n", sep)
    header = raw[:eol + 1].decode("ascii")
    dims = raw[sep:eol].split()
    assert dims[0] == b"-Y" and dims[2] == b"+X", dims
    h, w = int(dims[1]), int(dims[3])
    p = eol + 1
    out = np.zeros((h, w, 4), np.uint8)
    if not (8 <= w < 32768):
        out[:] = np.frombuffer(raw[p:p + h * w * 4], np.uint8).reshape(h, w, 4)
        return header, out
    for y in range(h):
        assert raw[p] == 2 and raw[p + 1] == 2 and ((raw[p + 2] << 8) | raw[p + 3]) == w, "not new-style RLE"
        p += 4
        for c in range(4):
            x = 0
            while x < w:
                n = raw[p]
                p += 1
                if n > 128:
                    n -= 128
                    out[y, x:x + n, c] = raw[p]
                    p += 1
                else:
                    out[y, x:x + n, c] = np.frombuffer(raw[p:p + n], np.uint8)
                    p += n
                x += n
            assert x == w
    assert p == len(raw), "trailing bytes"
    return header, out


def rgbe_to_float(rgbe):
    """Standard decode: (mantissa + 0.5) / 256 * 2^(e-128); zero exponent is black."""
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp((rgbe[..., :3].astype(np.float64) + 0.5) / 256.0, (e - 128)[..., None])
    f[e == 0] = 0.0
    return f.astype(np.float32)


def rgbe_lower_bound(rgbe):
    """Smallest float colour that the truncating encoder maps onto these bytes: mantissa / 256 * 2^(e-128)."""
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp(rgbe[..., :3].astype(np.float64) / 256.0, (e - 128)[..., None])
    f[e == 0] = 0.0
    return f.astype(np.float32)
