"""Per-kernel duration of the LAST n dispatches in a rocprofv3 --kernel-trace CSV (the timed launches of a probe; its first launches
settle the library's launch plan).   python tools/last_dispatch_stats.py <kernel_trace.csv> [n] [match]"""
import csv, sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
match = sys.argv[3] if len(sys.argv) > 3 else "trace_kernel"
print("Name,Calls,LastN,AverageNs_lastN,MinNs_lastN,MaxNs_lastN")
for name, v in sorted(rows.items()):
    if match in name:
        d = [x for _, x in sorted(v)[-n:]]
        print(f"\"{name}\",{len(v)},{len(d)},{sum(d) / len(d):.1f},{min(d)},{max(d)}")
