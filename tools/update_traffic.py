"""python tools/update_traffic.py <tag> [key]: turns gpurun_out/<tag>_traffic.json (tools/prof_traffic.sh) into the
entry of profiles/traffic.json that bench.py quotes as roofline.traffic, stamped with the profile it came from."""
import json
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "gl1_n1_720x720"
per_kernel = json.loads((root / "gpurun_out" / f"{tag}_traffic.json").read_text())
frame = sum(v["hbm_bytes_upper"] for k, v in per_kernel.items() if k.startswith("wf_") and k != "wf_cost_kernel")
path = root / "profiles" / "traffic.json"
table = json.loads(path.read_text()) if path.exists() else {}
table[key] = {"profile": f"profiles/{tag}_traffic.txt", "hbm_bytes_per_launch": frame,
              "note": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 summed over the frame's kernels, one frame per run", "per_kernel": per_kernel}
path.write_text(json.dumps(table, indent=1) + "\n")
print(key, frame / 1e9, "GB")
