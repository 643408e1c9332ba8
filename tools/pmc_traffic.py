"""Regenerates profiles/pmc_traffic.json on a GPU box: counters of the bench's traversal kernel, per launch, from SEPARATE
rocprofv3 --pmc passes (counters only, with --kernel-trace — never together with a sys / hip trace) of
`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe [bench args]`:

    FETCH_SIZE | WRITE_SIZE                                             bytes at the L2's fabric side (KB)
    TCP_TOTAL_CACHE_ACCESSES, TCP_TCC_READ_REQ, TCC_HIT, TCC_MISS       L1 lane requests, L1 -> L2 requests, L2 hits / misses
    SQ_* (waves, wave cycles, waits, VALU instructions / thread cycles) where the wave time goes, lane utilisation

Only the LAST 3 dispatches of the kernel (= the timed steps: bench.py's set-up passes try other launch plans through the same
kernel) are averaged. Stored with the sha1 of the traced kernel's instructions (tools/kernel_isa.py) so that bench.py can tell
whether the library it runs is the one that was traced.

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

PASSES = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU"],
    ["SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"],
]
STEPS = 3


def main():
    argv = sys.argv[1:]
    bench_args = argv[argv.index("--") + 1:] if "--" in argv else []
    out_path = argv[argv.index("--out") + 1] if "--out" in argv else os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
    out_path = os.path.abspath(out_path)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    work = os.path.join(os.path.dirname(out_path), "pmc_traffic_passes")
    env = dict(os.environ, TMPDIR="/tmp")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", "1", "--no-cpu-baseline", "--no-probe"] + bench_args
    # the bench line itself (kernel name, key fields) from an un-profiled run
    r = subprocess.run(base, capture_output=True, text=True, cwd=ROOT, timeout=900)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    kernel = line["roofline"]["kernel"]
    rays = line["config"]["rays_per_gpu_per_step"]
    want = kernel.replace(" ", "")
    # the profiled runs trace with the plan the un-profiled run settled on (under the counters' serialisation two close candidates can
    # swap places in the library's search, and the rows of the kernel asked for would be missing)
    plan = line["roofline"].get("launch_plan") or {}
    if plan:
        env.update(BVH_AMD_COOP=str(int(bool(plan.get("quad_cooperative_fetch")))), BVH_AMD_REFILL=str(plan.get("refill_threshold", 36)),
                   BVH_AMD_LEAF=str(plan.get("leaf_threshold", 12)))
        if not plan.get("reordered") and "--no-reorder" not in base:
            base = base + ["--no-reorder"]
    values = {}
    for i, counters in enumerate(PASSES):
        d = os.path.join(work, f"p{i}")
        subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + base,
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200, check=True)
        rows = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("bvh_amd::", "").replace("void ", "")
                if name.split("(")[0].replace(" ", "") == want:
                    rows.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        for c in counters:
            got = sorted(rows.get(c, []))[-STEPS:]
            if not got:
                raise SystemExit(f"no {c} rows for {kernel}")
            values[c] = sum(v for _, v in got) / len(got)
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
    doc["_doc"] = ("Counters per launch of the bench's traversal kernel from separate rocprofv3 --pmc passes (mean over the last 3 dispatches = the timed "
                   "steps; tools/pmc_traffic.py). FETCH_SIZE / WRITE_SIZE in KB at the L2's fabric side; FETCH_SIZE was calibrated on the record walk of "
                   "csrc/probe.hip over a 1 GiB table (a known byte count in this very access pattern): counter / known = 0.998 "
                   "(profiles/r03_fetch_calibration.json), so no correction applies to 64-byte record fetches. isa_sha1 = tools/kernel_isa.py of the traced "
                   "kernel: bench.py quotes the counts only for a library whose kernel hashes the same. Key: workload|quality|builder|traversal|rays per "
                   "launch|rays reordered by the call or traced as given.")
    v = values
    doc[key] = {"fetch_kb": round(v["FETCH_SIZE"], 1), "write_kb": round(v["WRITE_SIZE"], 1),
                "tcp_total_cache_accesses": round(v["TCP_TOTAL_CACHE_ACCESSES_sum"]), "tcp_tcc_read_req": round(v["TCP_TCC_READ_REQ_sum"]),
                "tcc_hit": round(v["TCC_HIT_sum"]), "tcc_miss": round(v["TCC_MISS_sum"]),
                "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4),
                "wave_time": {"waiting_at_s_waitcnt": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4), "issue_stalled": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4),
                              "issuing": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4)},
                "wave_instructions": {"valu": round(v["SQ_INSTS_VALU"]), "salu": round(v["SQ_INSTS_SALU"]), "vmem_read": round(v["SQ_INSTS_VMEM_RD"]),
                                      "lds": round(v["SQ_INSTS_LDS"]), "branch": round(v["SQ_INSTS_BRANCH"])},
                "raw": {k: round(x, 1) for k, x in v.items()},
                "kernel": kernel, "launch_plan": line["roofline"].get("launch_plan"),
                "isa_sha1": kernel_isa_hash(_lib.LIB_PATH, kernel), "kernel_ms_unprofiled": line["roofline"]["kernel_ms"], "round": 3}
    json.dump(doc, open(out_path, "w"), indent=2)
    print(json.dumps(doc[key]))


if __name__ == "__main__":
    main()
