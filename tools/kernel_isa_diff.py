"""Developer tool (CPU): which gfx950 kernels of two hipcc objects differ, instruction by instruction (branch offsets ignored).

    python tools/kernel_isa_diff.py old.o new.o [name-substring]

Used at the end of round 1 to show that adding the compact-record traversal kernels and splitting traverse.hip left all 106
`trace_kernel` instantiations instruction-identical (so the committed rocprof numbers still describe the shipped bench kernel):
build the old object from a git worktree of the earlier commit with the flags of bvh_amd/build.py, then compare.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def device_asm(obj):
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, "x.o")
    subprocess.check_call(["cp", obj, local])
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL)
    dev = [f for f in os.listdir(tmp) if "amdgcn" in f]
    assert dev, "no device code object in " + obj
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, dev[0])], capture_output=True, text=True, check=True).stdout


def functions(asm):
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        t = line.split("//")[0].strip()
        if t:
            out[cur].append(re.sub(r"(s_c?branch\S*)\s+\S+", r"\1 X", t))
    return out


def main():
    a, b = functions(device_asm(sys.argv[1])), functions(device_asm(sys.argv[2]))
    sel = sys.argv[3] if len(sys.argv) > 3 else ""
    same = diff = 0
    for k, v in a.items():
        if sel not in k:
            continue
        if k not in b:
            print("only in old:", k)
        elif v == b[k]:
            same += 1
        else:
            diff += 1
            print(f"DIFFERS: {k} ({len(v)} -> {len(b[k])} instructions)")
    new = [k for k in b if k not in a and sel in k]
    print(f"{same} identical, {diff} different, {len(new)} only in new")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
