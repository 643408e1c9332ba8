"""Developer probe: does a line-granular miss cost the memory system the same whether 64 or 128 bytes of the line are wanted? Dependent walks
over 64-byte records (one chain per quad, mode 2) and 128-byte records (one chain per octet, mode 5; lane 0 alone, mode 6), tables resident
in the L2s (2 MiB) and beyond them (1 GiB), G records/s.   python tools/wide_record_probe.py [out.txt]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import _lib

lib = _lib.load()


def table(n, words):
    perm = torch.randperm(n, device="cuda", dtype=torch.int64)
    t = torch.randint(0, 2 ** 31 - 1, (n, words), dtype=torch.int32, device="cuda")
    t[perm, 0] = torch.roll(perm, -1).to(torch.int32)
    return t


def rate(t, mode, bpc, steps):
    ms, recs = C.c_float(0), C.c_ulonglong(0)
    _lib.check(lib.bvh_amd_probe_record_walk_ex(t.data_ptr(), t.shape[0], steps, bpc, 3, mode, 64, C.byref(ms), C.byref(recs), None), "probe")
    return recs.value / (ms.value * 1e-3) / 1e9


out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for name, bytes_ in (("L2-resident 2 MiB", 2 << 20), ("beyond L2 1 GiB", 1 << 30)):
    t64, t128 = table(bytes_ // 64, 16), table(bytes_ // 128, 32)
    for bpc in (4, 8):
        line = (f"{name:18s} blocks/CU={bpc}: 64 B per quad {rate(t64, 2, bpc, 256):7.2f} | 64 B one lane of a quad {rate(t64, 3, bpc, 256):7.2f} | "
                f"128 B per octet {rate(t128, 5, bpc, 256):7.2f} | 128 B one lane of an octet {rate(t128, 6, bpc, 256):7.2f}  G records/s")
        print(line, flush=True)
        if out:
            out.write(line + "\n"); out.flush()
    del t64, t128
