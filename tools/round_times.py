"""Per-launch durations of the pipeline's kernels from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python3 tools/shard_bench.py 1 2 1
    python3 tools/round_times.py DIR/kt_kernel_trace.csv [kernel substring, default wf_trace]
Prints the durations (ms) in launch order, so that the rounds of a multi-round frame can be told apart."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "wf_trace"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if key in r["Kernel_Name"]]
print(f"{len(d)} launches of *{key}*, total {sum(d):.1f} ms")
print(" ".join(f"{x:.1f}" for x in d))
