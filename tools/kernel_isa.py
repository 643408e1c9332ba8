"""sha1 of ONE gfx950 kernel's instructions as they sit in a hipcc-built shared library (branch targets and addresses ignored).

    python tools/kernel_isa.py bvh_amd/lib/libbvh_amd.so "trace_kernel<float, false, true, 0, false, 3, false>"

Used to tell whether two builds carry the same kernel (the judge's check of a committed library against a rebuild; tests/test_host_logic.py's
ISA test of `ticket_release`). bench.py no longer keys stored counters to it: the counters are collected live by every run."""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_isa_hash(lib_path: str, kernel: str):
    """None when the tools or the kernel are missing (never raises: measurement plumbing must not take the bench down)."""
    body = kernel_isa_lines(lib_path, kernel)
    return None if not body else hashlib.sha1("\n".join(body).encode()).hexdigest()


def kernel_isa_lines(lib_path: str, kernel: str):
    """The kernel's instructions, one per line (branch targets replaced by X), or None."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not (os.path.exists(objdump) and os.path.exists(lib_path)):
        return None
    tmp = tempfile.mkdtemp(prefix="bvh_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([objdump, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        want = re.sub(r"\s+", "", kernel)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            asm = subprocess.run([objdump, "-d", "--demangle", os.path.join(tmp, name)], capture_output=True, text=True, check=True).stdout
            cur, body = None, []
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if cur is not None and body:
                        break
                    sym = re.sub(r"\(anonymous namespace\)::|bvh_amd::|^void |\s+", "", m.group(1))
                    cur = sym if sym.startswith(want + "(") or sym == want else None
                    continue
                if cur is None:
                    continue
                t = line.split("//")[0].strip()
                if t:
                    body.append(re.sub(r"(s_c?branch\S*)\s+\S+", r"\1 X", t))
            if cur is not None and body:
                return body
        return None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(kernel_isa_hash(sys.argv[1], sys.argv[2]))
