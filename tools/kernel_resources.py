"""Developer script: VGPR / SGPR / LDS / scratch of the kernels of the built library whose demangled name contains the argument.
    python tools/kernel_resources.py "trace_kernel_coop<float, false, true, 0, false>" """
import glob, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(pattern, lib=os.path.join(ROOT, "bvh_amd", "lib", "libbvh_amd.so")):
    out = []
    with tempfile.TemporaryDirectory(prefix="bvh_res_") as tmp:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, capture_output=True)
        for f in glob.glob(os.path.join(tmp, "lib.so.*amdgcn*")):
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
            for b in txt.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", b)
                if not name:
                    continue
                dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                if pattern not in dem:
                    continue
                g = lambda k: int((re.search(r"\." + k + r":\s+(\d+)", b) or [None, "-1"])[1])
                out.append({"kernel": dem, "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "lds": g("group_segment_fixed_size"),
                            "scratch": g("private_segment_fixed_size"), "vgpr_spills": g("vgpr_spill_count")})
    return out


if __name__ == "__main__":
    for r in kernel_resources(sys.argv[1] if len(sys.argv) > 1 else "trace_kernel"):
        print(f"{r['kernel'][:120]:120s} vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} scratch {r['scratch']:5d} spills {r['vgpr_spills']}")
