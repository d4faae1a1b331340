#!/usr/bin/env python3
"""Generates the data fixtures under tests/golden/ from the reference's committed renders.

Run in the build container only (needs /root/reference):

    python tests/golden/make_fixtures.py

Outputs (data only -- decoded pixel statistics and one output image of the reference):
  reference_hdr_stats.json   per-file mean RGB of the six .hdr renders, their header text, the SHA-256 of
                             the file and of its decoded RGBE bytes, and the 16x16-block mean image (45x45x3)
                             of every render whose settings the README records (gl_iter 0, 2, 2, 3); plus
                             "gl2_seed_band": the largest block-mean difference between the reference's two
                             gl_iter=2 renders (two seeds) -- the noise band of a block-mean comparison
                             Every file is re-encoded with the product's writer (crt_encode_hdr, a host
                             function) and must come back byte for byte, or the script stops.
  render0.hdr                byte copy of renders/render0.hdr, the reference's own output for README
                             row 0; pins the byte format of the .hdr writer
  ltc_tables_pin.json        SHA-256 of the 2 x 16384 float32 values of ltc_1 and ltc_2 as this script parses them
                             from the reference's ltc.hpp (line by line, four literals a line -- not the way
                             tools/extract_ltc.py reads them), plus a few spot values.  The product and the CPU
                             checker both read cudaraytracer_amd/data/ltc_tables.f32, so a wrong extraction would
                             be invisible to every GPU-against-checker comparison; tests/test_oracle_golden.py
                             holds that file to these digests.
"""
import hashlib
import json
import shutil
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))
import cudaraytracer_amd as crt  # noqa: E402  (crt_encode_hdr only: no GPU needed)
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
BLOCKS = ("renders/render0.hdr", "renders/render1.hdr", "renders/render2.hdr", "file75.hdr")


def ltc_pin():
    """Parses ltc_1 / ltc_2 (ltc.hpp:24-4124, 4127-8227) line by line and returns their digests."""
    lines = (REF / "ltc.hpp").read_text(errors="replace").splitlines()
    starts = [k for k, ln in enumerate(lines) if ln.startswith("__device__ float ltc_")]
    if len(starts) != 2:
        raise SystemExit("ltc.hpp: expected two __device__ float tables")
    tables = []
    for k in starts:
        vals = []
        for ln in lines[k + 1:]:
            t = ln.strip()
            if t.startswith("};"):
                break
            if not t or t[0] == "#" or t.startswith("static const"):
                continue
            toks = [x.strip() for x in t.rstrip(",").split(",")]
            if len(toks) != 4 or not all(x.endswith("f") for x in toks):
                raise SystemExit(f"ltc.hpp: unexpected table line {ln!r}")
            vals.extend(float(x[:-1]) for x in toks)
        if len(vals) != 64 * 64 * 4:
            raise SystemExit(f"ltc.hpp: table at line {k + 1} has {len(vals)} values")
        tables.append(np.array(vals, dtype=np.float64).astype("<f4"))
    t1, t2 = tables
    both = np.concatenate([t1, t2])
    return {"source": "ltc.hpp:24-4124 (ltc_1), ltc.hpp:4127-8227 (ltc_2); float32, row-major [y][x][channel]",
            "floats_per_table": int(t1.size),
            "ltc_1_sha256": hashlib.sha256(t1.tobytes()).hexdigest(),
            "ltc_2_sha256": hashlib.sha256(t2.tobytes()).hexdigest(),
            "both_sha256": hashlib.sha256(both.tobytes()).hexdigest(),
            "ltc_1_first8": [float(x) for x in t1[:8]], "ltc_1_last4": [float(x) for x in t1[-4:]],
            "ltc_2_first8": [float(x) for x in t2[:8]], "ltc_2_last4": [float(x) for x in t2[-4:]],
            "sum_float64": [float(t1.astype(np.float64).sum()), float(t2.astype(np.float64).sum())]}


def main():
    (HERE / "ltc_tables_pin.json").write_text(json.dumps(ltc_pin(), indent=1))
    stats = {}
    for rel, note in FILES.items():
        raw = (REF / rel).read_bytes()
        header, rgbe = decode_hdr(raw)
        img = rgbe_to_float(rgbe)
        entry = {"note": note, "header": header, "height": int(img.shape[0]), "width": int(img.shape[1]),
                 "bytes": len(raw), "mean_rgb": [float(x) for x in img.reshape(-1, 3).astype(np.float64).mean(0)]}
        entry["file_sha256"] = hashlib.sha256(raw).hexdigest()
        entry["rgbe_sha256"] = hashlib.sha256(np.ascontiguousarray(rgbe).tobytes()).hexdigest()
        # the writer under test reproduces the reference's file from its decoded (mid-bucket) colours
        again = crt.encode_hdr(img)
        if again != raw:
            raise SystemExit(f"{rel}: re-encoding differs from the reference's bytes")
        entry["reencode_identical"] = True
        if rel in BLOCKS:
            h, w, _ = img.shape
            blocks = img.reshape(h // 16, 16, w // 16, 16, 3).astype(np.float64).mean((1, 3))
            entry["block16_mean"] = np.round(blocks, 7).tolist()
        stats[rel] = entry
    a = np.array(stats["renders/render1.hdr"]["block16_mean"])
    b = np.array(stats["file75.hdr"]["block16_mean"])
    stats["gl2_seed_band"] = {"note": "renders/render1.hdr vs file75.hdr, 16x16 block means",
                              "max_abs": float(np.abs(a - b).max()), "mean_abs": float(np.abs(a - b).mean()),
                              "max_rel_mean_rgb": float(np.abs(np.array(stats["renders/render1.hdr"]["mean_rgb"]) /
                                                               np.array(stats["file75.hdr"]["mean_rgb"]) - 1).max())}
    (HERE / "reference_hdr_stats.json").write_text(json.dumps(stats, indent=1))
    shutil.copyfile(REF / "renders/render0.hdr", HERE / "render0.hdr")
    (HERE / "render0.hdr").chmod(0o644)
    print("wrote", HERE / "reference_hdr_stats.json", "and render0.hdr")
    for k, v in stats.items():
        print(k, v.get("mean_rgb", v))


if __name__ == "__main__":
    main()
