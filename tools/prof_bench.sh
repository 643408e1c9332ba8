# rocprofv3 kernel trace of the default bench command (N = 1): gpurun_out/r06_bench_soup1m_kernel_stats.csv + the line it printed
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-pmc > gpurun_out/r06_bench_under_rocprof.json 2> gpurun_out/r06_bench_under_rocprof.err
db=$(ls gpurun_out/prof_bench/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" gpurun_out/r06_bench_soup1m_kernel_stats.csv | head -6 | cut -c1-200; else echo "no rocpd database"; ls gpurun_out/prof_bench | head; fi
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_under_rocprof.json"))
print("value", d["value"], "kernel_ms", d["roofline"]["kernel_ms"], d["roofline"]["kernel"])
PY
