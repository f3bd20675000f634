import sqlite3, collections, re, sys
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name, start, end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if ('jk_dense_kernel' in r[0] or 'jk_sym_kernel' in r[0])]
i0, i1 = idx[-11], idx[-1]
def short(n):
    n=n.replace('(anonymous namespace)::','').replace('void ','')
    m=re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)',n)
    return (m.group(1) if m else n)[:70]
print("cycle period ms", (rows[i1][1]-rows[i0][1])/10/1e6)
agg=collections.defaultdict(lambda:[0,0.0]); busy=0
for r in rows[i0:i1]:
    n=short(r[0]); agg[n][0]+=1; agg[n][1]+=(r[2]-r[1])/1e6; busy+=(r[2]-r[1])/1e6
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"{n:72s} {c/10:6.1f}/cyc {t/10*1000:8.1f} us/cyc")
print("busy ms/cycle", busy/10, "kernels/cycle", (i1-i0)/10)
if len(sys.argv)>2:
    j0=idx[-2]; prev=None
    for r in rows[j0:idx[-1]+1]:
        gap = (r[1]-prev)/1e3 if prev else 0
        print(f"gap {gap:7.1f} us  dur {(r[2]-r[1])/1e3:7.1f} us  {short(r[0])}")
        prev=r[2]
