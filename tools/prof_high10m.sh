# rocprofv3 kernel trace of three 10M-triangle High builds (tools/time_high_sizes.py); the summary goes to gpurun_out/r06_high10m_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_high10m -o high10m -- python tools/time_high_sizes.py 10000000 > gpurun_out/prof_high10m.log 2>&1
db=$(ls gpurun_out/prof_high10m/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" gpurun_out/r06_high10m_kernel_stats.csv | head -12 | cut -c1-160; else echo "no rocpd database"; fi
