"""Developer probe: do the L2's and the fabric's service times ADD for one CU (one shared resource: the lines its L1 can keep outstanding)
or overlap? Pure walks over a 2 MiB and a 1 GiB table against one chain per lane that alternates between the two (csrc/probe.hip:
k_record_walk_mixed), per-lane and quad-cooperative fetch, several occupancies.   python tools/mixed_walk_probe.py [out.txt]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import _lib

lib = _lib.load()


def table(n):
    perm = torch.randperm(n, device="cuda", dtype=torch.int64)
    t = torch.randint(0, 2 ** 31 - 1, (n, 16), dtype=torch.int32, device="cuda")
    t[perm, 0] = torch.roll(perm, -1).to(torch.int32)
    return t


def pure(t, mode, bpc, steps=256):
    ms, recs = C.c_float(0), C.c_ulonglong(0)
    _lib.check(lib.bvh_amd_probe_record_walk_ex(t.data_ptr(), t.shape[0], steps, bpc, 3, mode, 64, C.byref(ms), C.byref(recs), None), "probe")
    return recs.value / (ms.value * 1e-3) / 1e9


def mixed(ts, tb, coop, bpc, steps=256):
    ms, recs = C.c_float(0), C.c_ulonglong(0)
    _lib.check(lib.bvh_amd_probe_mixed_walk(ts.data_ptr(), ts.shape[0], tb.data_ptr(), tb.shape[0], steps, bpc, 3, coop, C.byref(ms), C.byref(recs), None), "mixed")
    return recs.value / (ms.value * 1e-3) / 1e9


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None

    def emit(s):
        print(s, flush=True)
        if out:
            out.write(s + "\n"); out.flush()
    ts, tb = table(32768), table(16777216)
    emit("# G record fetches / s; 'add' = 2 / (1 / R_L2 + 1 / R_fabric): the times of the two levels add; 'overlap' = 2 x R_fabric: only the slower level's throughput counts")
    for coop in (0, 1):
        mode = 4 if coop else 0
        for bpc in (2, 4, 7, 8):
            r2, rf = pure(ts, mode, bpc, 512), pure(tb, mode, bpc)
            m = mixed(ts, tb, coop, bpc)
            emit(f"{'quad-cooperative' if coop else 'per lane        '} blocks/CU={bpc}: L2-resident {r2:7.2f}  beyond-L2 {rf:6.2f}  alternating {m:7.2f}   add -> {2 / (1 / r2 + 1 / rf):7.2f}   overlap -> {min(2 * rf, 2 * r2):7.2f}")


if __name__ == "__main__":
    main()
