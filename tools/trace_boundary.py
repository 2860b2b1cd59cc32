import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "ms_fwd_adj_kernel" in r["Kernel_Name"]]
st=[int(rows[i]["Start_Timestamp"]) for i in idx]
d=[(b-a)/1e3 for a,b in zip(st[:-1],st[1:])]
print(" ".join("%.0f"%v for v in d[-70:]))
cands=[k for k in range(len(st)-40,len(st)-1) if st[k+1]-st[k]>2.4e6]
k=cands[0]
a,b=idx[k-1],idx[k+1]
t0=int(rows[a]["Start_Timestamp"]); prev=t0
for r in rows[a:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"]; n=n[:n.index("(")] if "(" in n else n
    if "PRIM" in n: continue
    print("%9.1f us  dur %8.1f  gap %7.1f  q%-3s %s"%((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,r.get("Queue_Id","?"),n[-50:]))
    prev=max(prev,e)
