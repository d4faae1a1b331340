"""The .hdr writer against the reference's own output file (stbi_write_hdr at main.cu:856)."""
from pathlib import Path

import numpy as np

import cudaraytracer_amd as crt
from hdr_codec import decode_hdr, rgbe_lower_bound, rgbe_to_float

GOLD = Path(__file__).resolve().parent / "golden"


def test_reencoding_the_reference_render_reproduces_its_bytes():
    raw = (GOLD / "render0.hdr").read_bytes()
    header, rgbe = decode_hdr(raw)
    assert header == ("#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\n"
                      "EXPOSURE=          1.0000000000000\n\n-Y 720 +X 720\n")
    rgb = rgbe_to_float(rgbe)  # mid-bucket colours: truncation maps them back onto the same bytes
    img = np.concatenate([rgb, np.ones_like(rgb[..., :1])], axis=-1)  # comp = 4, alpha ignored
    assert crt.encode_hdr(img) == raw
    assert crt.encode_hdr(rgb) == raw  # comp = 3


def test_mantissa_is_truncated_not_rounded():
    _, rgbe = decode_hdr((GOLD / "render0.hdr").read_bytes())
    lo = rgbe_lower_bound(rgbe)
    _, again = decode_hdr(crt.encode_hdr(lo.astype(np.float32)))
    # the bucket's lower edge must land in the same bucket (or, by float rounding of the scale, one below)
    d = again[..., :3].astype(int) - rgbe[..., :3].astype(int)
    assert d.max() <= 0 and d.min() >= -1
    assert (again[..., 3] == rgbe[..., 3]).mean() > 0.999


def test_narrow_images_are_written_flat(tmp_path):
    img = np.linspace(0, 2, 5 * 3 * 3, dtype=np.float32).reshape(3, 5, 3)
    img[0, 0] = 0.0
    raw = crt.encode_hdr(img)
    head = b"#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y 3 +X 5\n"
    assert raw.startswith(head) and len(raw) == len(head) + 3 * 5 * 4
    _, rgbe = decode_hdr(raw)
    assert np.abs(rgbe_to_float(rgbe) - img).max() <= np.abs(img).max() / 128
    assert rgbe[0, 0].tolist() == [0, 0, 0, 0]  # black pixel
    path = tmp_path / "file.hdr"
    crt.write_hdr(path, img)
    assert path.read_bytes() == raw


def test_runs_literals_and_long_rows_roundtrip():
    rng = np.random.default_rng(1)
    w = 300
    row = np.repeat(0.5 + 0.5 * rng.random(30, dtype=np.float32), 10)  # runs of 10; max in [0.5,1): one exponent
    img = np.stack([np.stack([row, 0.5 * rng.random(w, dtype=np.float32), np.full(w, 0.25, np.float32)], -1)] * 2)
    raw = crt.encode_hdr(img)
    _, rgbe = decode_hdr(raw)
    assert np.abs(rgbe_to_float(rgbe) - img).max() < 1 / 128
    # a constant plane of 300 bytes needs three run packets (127 + 127 + 46)
    assert bytes([128 + 127, rgbe[0, 0, 2], 128 + 127, rgbe[0, 0, 2], 128 + 46, rgbe[0, 0, 2]]) in raw


def test_rgba8_preview_matches_the_reference_conversion(tmp_path):
    """main.cu:816-818: static_cast<uint8_t>(std::clamp(c * 255.0f, 0.f, 255.f)), then a PPM of the RGB part."""
    import cudaraytracer_amd as crt
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.2, 1.3, size=(9, 13, 4)).astype(np.float32)
    img[0, 0] = [0.0, 1.0, 0.5, 254.999 / 255.0]
    img[0, 1] = [np.nan, np.inf, -np.inf, 1e-9]
    got = crt.to_rgba8(img)
    with np.errstate(invalid="ignore"):
        want = np.clip(img * np.float32(255.0), 0.0, 255.0)
    want = np.where(np.isnan(want), 0.0, want).astype(np.uint8)  # truncation
    assert np.array_equal(got, want)
    assert list(got[0, 0]) == [0, 255, 127, 254]
    path = tmp_path / "preview.ppm"
    crt.write_ppm(path, got)
    raw = path.read_bytes()
    header = b"P6\n13 9\n255\n"
    assert raw.startswith(header)
    assert np.array_equal(np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(9, 13, 3), got[:, :, :3])
