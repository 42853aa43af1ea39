for m in 0 1; do
MTLORA_NTD=$m MTLORA_PROF_DUMP=gpurun_out/dump_c4_$m.csv python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-eager-gpu --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NTD=$m', d['value'], d['ms_per_step'])"
done
python - <<'PY'
import collections
def load(f):
    g=collections.OrderedDict()
    for l in open(f):
        k,b,s8,fl,ms,tag=l.rstrip("\n").split(",",5)
        a=g.setdefault((k,tag),[0,0.0]); a[0]+=1; a[1]+=float(ms)
    return g
a=load("gpurun_out/dump_c4_0.csv"); b=load("gpurun_out/dump_c4_1.csv")
rows=[]
for key,(c,t) in a.items():
    if key in b and key[0].startswith("k_nt"):
        rows.append((t/5-b[key][1]/5, key, c/5, 1e3*t/c, 1e3*b[key][1]/b[key][0]))
for d,key,n,u0,u1 in sorted(rows, key=lambda r:-abs(r[0]))[:25]: print(f"{key[0]:20s} {key[1]:46s} n {n:4.1f}  {u0:7.1f} -> {u1:7.1f} us   d {d:+.3f} ms/step")
PY
