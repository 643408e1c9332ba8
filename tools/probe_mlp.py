"""Developer probe (VERDICT r3 "Next 5"): are the records/s ceilings bench.py prices the traversal kernel against a property of the
memory system, or of a probe that keeps ONE dependent fetch in flight per lane? Dependent walks over random 64-byte records with 1, 2
and 4 INDEPENDENT chains per lane (csrc/probe.hip modes 0 / 7 / 8: per-lane loads; 4 / 9: quad-cooperative fetch with 1 / 2 chains per
lane), at an L2-resident table, a table around the L2s and two beyond them, 4 and 8 blocks of 256 lanes per CU, all lanes active and 28.
    python tools/probe_mlp.py [out.txt]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import _lib
from tcp_probe import table, run

NAMES = {0: "per-lane, 1 chain/lane", 7: "per-lane, 2 chains/lane", 8: "per-lane, 4 chains/lane", 4: "quad-coop, 1 chain/lane", 9: "quad-coop, 2 chains/lane"}


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None

    def emit(s):
        print(s, flush=True)
        if out:
            out.write(s + "\n"); out.flush()
    emit(f"# {torch.cuda.get_device_name(0)}; G records/s of dependent walks over random 64-byte records (records in flight per lane = chains per lane)")
    for name, n in (("16KiB(L1)", 256), ("2MiB(L2)", 32768), ("24MiB(~L2s)", 393216), ("108MiB(MALL)", 1769472), ("1GiB(HBM)", 16777216)):
        t = table(n)
        for bpc in (2, 4, 8):
            for active in (64, 28):
                cells = []
                for mode in (0, 7, 8, 4, 9):
                    ms, recs = run(t, mode, active, bpc, steps=512 if n <= 32768 else 192)
                    cells.append(f"{NAMES[mode]}: {recs / (ms * 1e-3) / 1e9:7.2f}")
                emit(f"table={name:13s} blocks/CU={bpc} active={active:2d} | " + " | ".join(cells))
        del t


if __name__ == "__main__":
    main()
