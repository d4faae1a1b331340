#!/usr/bin/env python3
"""Generates the data fixtures under tests/golden/ from the reference's committed renders.

Run in the build container only (needs /root/reference):

    python tests/golden/make_fixtures.py

Outputs (data only -- decoded pixel statistics and one output image of the reference):
  reference_hdr_stats.json   per-file mean RGB of the six .hdr renders, their header text and, for
                             renders/render0.hdr (gl_iter=0), the 16x16-block mean image (45x45x3)
  render0.hdr                byte copy of renders/render0.hdr, the reference's own output for README
                             row 0; pins the byte format of the .hdr writer
"""
import json
import shutil
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from hdr_codec import decode_hdr, rgbe_to_float  # noqa: E402

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
FILES = {
    "renders/render0.hdr": "gl_iter=0 (README.md:104-108)",
    "renders/render1.hdr": "gl_iter=2 (README.md:110-114)",
    "renders/render2.hdr": "gl_iter=3 (README.md:116-120)",
    "file.hdr": "identical to renders/render2.hdr",
    "file75.hdr": "gl_iter=2, ind_rays=75, another seed (settings not recorded)",
    "file5.hdr": "settings not recorded",
}


def main():
    stats = {}
    for rel, note in FILES.items():
        raw = (REF / rel).read_bytes()
        header, rgbe = decode_hdr(raw)
        img = rgbe_to_float(rgbe)
        entry = {"note": note, "header": header, "height": int(img.shape[0]), "width": int(img.shape[1]),
                 "bytes": len(raw), "mean_rgb": [float(x) for x in img.reshape(-1, 3).mean(0)]}
        if rel == "renders/render0.hdr":
            h, w, _ = img.shape
            blocks = img.reshape(h // 16, 16, w // 16, 16, 3).mean((1, 3))
            entry["block16_mean"] = np.round(blocks, 6).tolist()
        stats[rel] = entry
    (HERE / "reference_hdr_stats.json").write_text(json.dumps(stats, indent=1))
    shutil.copyfile(REF / "renders/render0.hdr", HERE / "render0.hdr")
    (HERE / "render0.hdr").chmod(0o644)
    print("wrote", HERE / "reference_hdr_stats.json", "and render0.hdr")
    for k, v in stats.items():
        print(k, v["mean_rgb"])


if __name__ == "__main__":
    main()
