"""FETCH_SIZE calibration on a KNOWN byte count in the traversal's own access pattern (MI355X_MICROARCH.md, HBM section: "calibrate
on a known byte count in your own access pattern before trusting an absolute"): the record walk over a 1 GiB table (every 64-byte
record fetch misses the 32 MiB of L2) moves records x 64 bytes through the L2's fabric side.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/fetch_calibration.py
prints the bytes the launch must have fetched; tools/fetch_calibration.py --read <dir> divides the counter by it."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def walk():
    import ctypes as C
    import torch
    from bvh_amd import _lib
    n = 16777216                                               # 1 GiB of 64-byte records
    perm = torch.randperm(n, device="cuda", dtype=torch.int64)
    t = torch.randint(0, 2 ** 31 - 1, (n, 16), dtype=torch.int32, device="cuda")
    t[perm, 0] = torch.roll(perm, -1).to(torch.int32)
    del perm
    ms, recs = C.c_float(0), C.c_ulonglong(0)
    _lib.check(_lib.load().bvh_amd_probe_record_walk_ex(t.data_ptr(), n, 64, 7, 3, 0, 64, C.byref(ms), C.byref(recs), None), "probe")
    print(json.dumps({"records_per_launch": recs.value, "bytes_per_launch": recs.value * 64, "ms": ms.value,
                      "note": "each of the 4 launches (1 warm-up + 3) walks this many records; the first touch of a chain start may hit lines a neighbour fetched: < 1 % here"}))


def read(d):
    vals = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "k_record_walk" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
                vals.append(float(row["Counter_Value"]))
    return vals


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        v = read(sys.argv[2])
        known = float(sys.argv[3]) if len(sys.argv) > 3 else None
        mean_kb = sum(v) / max(1, len(v))
        print(json.dumps({"fetch_size_kb_per_launch": mean_kb, "launches": len(v), "known_bytes": known,
                          "factor_counter_over_known": None if not known else mean_kb * 1024.0 / known}))
    else:
        walk()
