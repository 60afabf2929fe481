import sqlite3,glob,collections,sys
db=sorted(glob.glob(sys.argv[1]+"/**/*.db",recursive=True))[-1]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows=c.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%attn%' order by d.start").fetchall()
agg=collections.defaultdict(list)
import re
for n,dt in rows:
    m=re.search(r"attn_\w+?_kernel", n); agg[m.group(0) if m else n[:40]].append(dt)
print(sys.argv[2], " ".join(f"{k}: {sorted(v)[len(v)//2]/1e3:.1f} us" for k,v in agg.items()))
