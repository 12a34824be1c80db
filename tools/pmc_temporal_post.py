import collections, glob, sqlite3, sys
out=sys.argv[1]
rows=collections.OrderedDict()
for run in ("run1","run2"):
    db=sqlite3.connect(glob.glob(f"{out}/{run}/**/*.db",recursive=True)[0])
    tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view="counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counters_collection" in t][0]
    cols=[r[1] for r in db.execute(f"pragma table_info('{view}')")]
    ci={c:i for i,c in enumerate(cols)}
    nm="kernel_name" if "kernel_name" in ci else "name"
    per=collections.OrderedDict()
    for r in db.execute(f"select * from {view} order by dispatch_id"):
        kn=str(r[ci[nm]])
        if "temporal_layer" not in kn: continue
        key="13 waves (WMODE 5)" if "layer13" in kn else ("8 waves (WMODE 4) <%s>"%kn.split("temporal_layer16_kernel<")[1][0] if "layer16" in kn else "32x32 (WMODE 3)")
        gs = r[ci["grid_size"]] if "grid_size" in ci else 0
        d=per.setdefault((key,gs,r[ci["dispatch_id"]]),{})
        d[r[ci["counter_name"]]]=d.get(r[ci["counter_name"]],0.0)+float(r[ci["value"]])
    for (key,gs,_),d in per.items():
        rr=rows.setdefault((key,gs),collections.defaultdict(list))
        for c,v in d.items(): rr[c].append(v)
names=sorted({c for d in rows.values() for c in d})
keys=sorted(rows.keys(), key=lambda k:(-k[1] if False else 0, k[0]))
print("# SQ counters per launch of the fused 64-channel temporal layer (tools/pmc_temporal_layer.sh: two rocprofv3 --pmc passes over tools/bench_temporal_layer.py; medians; sums over the chip; `grid` = threads of the launch: 4096 or 1024 pixel columns x workgroup size)\n")
print("| counter | "+" | ".join(f"{k[0]}, grid {k[1]}" for k in keys)+" |")
print("|---|"+"---|"*len(keys))
for c in names:
    vals=[]
    for k in keys:
        v=sorted(rows[k].get(c,[0])); vals.append(f"{v[len(v)//2]:.4g}")
    print(f"| {c} | "+" | ".join(vals)+" |")
