import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    return n.replace('bvh_amd::(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
d = collections.defaultdict(list)
for r in rows[-400:]:
    d[short(r['Kernel_Name'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k:46s} n={len(v):4d} median {v[len(v)//2]:9.1f} us  max {v[-1]:9.1f}")
