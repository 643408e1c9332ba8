# SQ counters of the pair kernel and the wide kernel on the soup (tools/wide_check.py, soup only), one rocprofv3 --pmc pass
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
WIDE_CHECK_ONLY=soup_1m timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/wide_pmc -- python tools/wide_check.py 22 > gpurun_out/wide_pmc.log 2>&1
f=$(find gpurun_out/wide_pmc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "trace_kernel" in n and "plan_search" not in n:
        rows[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in rows.items():
    print(k, {c: round(sum(x[-3:]) / len(x[-3:])) for c, x in v.items()})
PY
