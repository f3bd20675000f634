import csv,re,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r:r[1])
idx=[i for i,r in enumerate(rows) if re.search(r"jk_(s4|m4|m8|sym|dense)_kernel",r[0])]
def short(n):
    return n.replace("(anonymous namespace)::","").replace("void ","")[:50]
c=len(idx)-3
prev=None
for r in rows[idx[c]:idx[c+1]+1]:
    gap=(r[1]-prev)/1e3 if prev else 0
    print(f"  gap {gap:6.1f} dur {(r[2]-r[1])/1e3:7.1f} {short(r[0])}")
    prev=r[2]
print("period", (rows[idx[c+1]][1]-rows[idx[c]][1])/1e3)
