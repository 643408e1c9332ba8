"""Regenerates profiles/pmc_traffic.json on a GPU box: the HBM-side bytes per launch of the bench's traversal kernel, from two
SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, with --kernel-trace — never together with a sys / hip
trace) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe [bench args]`, plus the sha1 of the traced kernel's
instructions (tools/kernel_isa.py) so that bench.py can tell whether the library it runs is the one that was traced.

    python tools/pmc_traffic.py [--out gpurun_out/pmc_traffic.json] [-- bench args ...]        (then copy the file to profiles/)
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    argv = sys.argv[1:]
    bench_args = argv[argv.index("--") + 1:] if "--" in argv else []
    out_path = argv[argv.index("--out") + 1] if "--out" in argv else os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
    out_path = os.path.abspath(out_path)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    work = os.path.join(os.path.dirname(out_path), "pmc_traffic_passes")
    env = dict(os.environ, TMPDIR="/tmp")
    # the bench line itself (kernel name, key fields) from an un-profiled run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-probe"] + bench_args,
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    kernel = line["roofline"]["kernel"]
    rays = line["config"]["rays_per_gpu_per_step"]
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(work, counter)
        subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-probe"] + bench_args,
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200, check=True)
        vals = []
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("bvh_amd::", "").replace("void ", "")
                if name.split("(")[0].replace(" ", "") == kernel.replace(" ", "") and row["Counter_Name"] == counter:
                    vals.append(float(row["Counter_Value"]))
        if not vals:
            raise SystemExit(f"no {counter} rows for {kernel}")
        sums[counter] = (sum(vals) / len(vals), len(vals))
    from kernel_isa import kernel_isa_hash
    from bvh_amd import _lib
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="soup_1m"); ap.add_argument("--quality", default="high")
    ap.add_argument("--serial-builder", action="store_true"); ap.add_argument("--fast", action="store_true")
    a, _ = ap.parse_known_args(bench_args)
    order = "reordered" if str(line["roofline"].get("ray_reordering", "off")).startswith("on") else "as_given"
    key = f"{a.workload}|{a.quality}|{'serial' if a.serial_builder else 'pool'}|{'fast' if a.fast else 'robust'}|{rays}|{order}"
    doc = {}
    if os.path.exists(out_path):
        doc = json.load(open(out_path))
    doc["_doc"] = ("HBM-side traffic per launch of the bench's traversal kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in KB, "
                   "mean over the kernel's dispatches; tools/pmc_traffic.py). Requests are 64-byte record sectors, so the guide's 2x wide-stream "
                   "correction does not apply. isa_sha1 = tools/kernel_isa.py of the traced kernel: bench.py quotes the counts only for a library "
                   "whose kernel hashes the same. Key: workload|quality|builder|traversal|rays per launch|rays reordered by the call or traced as given.")
    doc[key] = {"fetch_kb": round(sums["FETCH_SIZE"][0], 1), "write_kb": round(sums["WRITE_SIZE"][0], 1),
                "dispatches": [sums["FETCH_SIZE"][1], sums["WRITE_SIZE"][1]], "kernel": kernel,
                "isa_sha1": kernel_isa_hash(_lib.LIB_PATH, kernel), "kernel_ms_unprofiled": line["roofline"]["kernel_ms"], "round": 2}
    json.dump(doc, open(out_path, "w"), indent=2)
    print(json.dumps(doc[key]))


if __name__ == "__main__":
    main()
