#!/usr/bin/env python3
"""Extract the two 64x64x4 LTC lookup tables as raw little-endian float32.

The tables are numeric data (the public GGX fit of Heitz et al., "Real-Time
Polygonal-Light Shading with Linearly Transformed Cosines", distributed with
selfshadow/ltc_code); the reference carries them as the initialisers of
`ltc_1` and `ltc_2` (ltc.hpp:24-4124 and ltc.hpp:4127-8227).  This script reads
only the numeric literals and writes

    cudaraytracer_amd/data/ltc_tables.f32   (2 * 64*64*4 floats = 131072 bytes)

with ltc_1 first, ltc_2 second, row-major [y][x][channel] exactly as indexed by
Texture::getPixelColor (Texture.hpp:211-237).  Run in the build container only:

    python tools/extract_ltc.py /root/reference/ltc.hpp
"""
import re
import sys
from pathlib import Path

import numpy as np

N = 64 * 64 * 4


def parse_tables(text: str):
    tables = []
    for name in ("ltc_1", "ltc_2"):
        m = re.search(r"__device__\s+float\s+%s\[\]\s*=\s*\{" % name, text)
        if not m:
            raise SystemExit(f"table {name} not found")
        end = text.index("};", m.end())
        body = text[m.end():end]
        # drop the host-side alternative declaration that sits between #else/#endif
        body = re.sub(r"#else.*?#endif", "", body, flags=re.S)
        lits = re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?f", body)
        vals = np.array([np.float32(s[:-1]) for s in lits], dtype=np.float32)
        if vals.size != N:
            raise SystemExit(f"{name}: expected {N} values, got {vals.size}")
        tables.append(vals)
    return tables


def main():
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/ltc.hpp")
    out = Path(__file__).resolve().parent.parent / "cudaraytracer_amd" / "data" / "ltc_tables.f32"
    t1, t2 = parse_tables(src.read_text(errors="replace"))
    out.parent.mkdir(parents=True, exist_ok=True)
    np.concatenate([t1, t2]).astype("<f4").tofile(out)
    print(f"wrote {out} ({out.stat().st_size} bytes); "
          f"ltc_1[0:4]={t1[:4]} ltc_1[-4:]={t1[-4:]} ltc_2[4:8]={t2[4:8]}")


if __name__ == "__main__":
    main()
