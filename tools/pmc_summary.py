"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one directory per pass)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, match="trace_kernel", last=0):
    """last > 0: only the last `last` dispatches of every kernel count (the timed launches of a probe whose first launches settle
    the library's launch plan with other plans through the same kernel)."""
    out = []
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        rows = defaultdict(list)
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            rows[(name.split("(")[0], row["Counter_Name"])].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        for (k, c), vals in sorted(rows.items()):
            if match in k:
                vals = sorted(vals)[-last:] if last > 0 else vals
                out.append(f"{os.path.relpath(path, root).split(os.sep)[0]},\"{k}\",{c},{len(vals)},{sum(v for _, v in vals) / len(vals):.1f}")
    print("pass,kernel,counter,dispatches,avg_per_dispatch")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *(int(a) for a in sys.argv[3:4]))
