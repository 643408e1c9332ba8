"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one directory per pass)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, match="trace_kernel"):
    out = []
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            key = (name.split("(")[0], row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
        for (k, c), (s, n) in sorted(acc.items()):
            if match in k:
                out.append(f"{os.path.relpath(path, root).split(os.sep)[0]},\"{k}\",{c},{n},{s / n:.1f}")
    print("pass,kernel,counter,dispatches,avg_per_dispatch")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
