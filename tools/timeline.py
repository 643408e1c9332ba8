"""Developer tool: kernel timeline (start, gap to the previous kernel's end, duration) of the LAST build in a rocprofv3 --kernel-trace CSV
of tools/build_profile.py (the last tri_bounds_kernel marks its start).   python tools/timeline.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tri_bounds_kernel' in r['Kernel_Name']]
seg = rows[idx[-1]:]
t0 = int(seg[0]['Start_Timestamp']); prev_end = t0; tot_gap = 0; tot_k = 0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = s - prev_end
    name = r['Kernel_Name'].replace('void bvh_amd::(anonymous namespace)::', '').replace('bvh_amd::(anonymous namespace)::', '').replace('void bvh_amd::bld::', '').split('(')[0][:44]
    if max(gap, 0) / 1e3 >= min_us or (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:9.1f}us gap={gap / 1e3:7.1f} dur={(e - s) / 1e3:8.1f} {name}")
    tot_gap += max(gap, 0); tot_k += e - s; prev_end = max(prev_end, e)
print('total', (prev_end - t0) / 1e3, 'us; kernels', tot_k / 1e3, 'gaps', tot_gap / 1e3, 'launches', len(seg))
