"""Developer probe: the record-walk microbenchmark matrix (csrc/probe.hip) — what the L1 (TCP), the L2 and the fabric side give
the traversal's access pattern, per-lane loads against quad-cooperative loads, full and partially active waves.
    python tools/tcp_probe.py [out.txt]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import _lib


def table(n):
    perm = torch.randperm(n, device="cuda", dtype=torch.int64)
    t = torch.randint(0, 2 ** 31 - 1, (n, 16), dtype=torch.int32, device="cuda")
    t[perm, 0] = torch.roll(perm, -1).to(torch.int32)
    return t


def run(t, mode, active, bpc, steps=256, reps=3):
    lib = _lib.load()
    ms, recs = C.c_float(0), C.c_ulonglong(0)
    _lib.check(lib.bvh_amd_probe_record_walk_ex(t.data_ptr(), t.shape[0], steps, bpc, reps, mode, active, C.byref(ms), C.byref(recs), None), "probe")
    return ms.value, recs.value


MODES = ((0, 64), (0, 48), (0, 28), (0, 16), (4, 64), (4, 48), (4, 28), (4, 16), (1, 64), (1, 28), (2, 64), (3, 64)) if not os.environ.get("PROBE_MODES") else \
    tuple(tuple(int(x) for x in m.split(":")) for m in os.environ["PROBE_MODES"].split(","))


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    clk = 2.4e9

    def emit(s):
        print(s, flush=True)
        if out:
            out.write(s + "\n"); out.flush()
    emit(f"# {torch.cuda.get_device_name(0)}, {cus} CUs; G records/s; lane-req/clk/CU = what mode 0 asks of the L1 (4 requests per record) at 2.4 GHz")
    for name, n in (("8KiB(L1)", 128), ("16KiB(L1)", 256), ("2MiB(L2)", 32768), ("24MiB(L2s)", 393216), ("108MiB", 1769472), ("1GiB", 16777216)):
        t = table(n)
        for bpc in (7,):
            for mode, active in MODES:
                ms, recs = run(t, mode, active, bpc, steps=512 if n <= 32768 else 256)
                rate = recs / (ms * 1e-3)
                req = {0: 4 * rate, 1: rate, 2: rate, 3: 4 * rate, 4: rate}[mode] / cus / clk
                emit(f"table={name:11s} blocks/CU={bpc} mode={mode} active={active:2d}: {ms:8.4f} ms {rate / 1e9:8.2f} Grec/s  {rate * 64 / 1e12:6.2f} TB/s  L1 line-req/clk/CU={req:5.2f}")
        del t


if __name__ == "__main__":
    main()
