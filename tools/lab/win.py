import csv,re,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r.get("Queue_Id","?"),re.sub(r"^void cwt::","",r["Kernel_Name"]).split("(")[0]))
rows.sort()
t0=rows[0][0]
pr=[i for i,r in enumerate(rows) if r[3].startswith("k_poly_rows")]
k=int(sys.argv[2]) if len(sys.argv)>2 else 10
a=(rows[pr[k]][0]-t0)/1e3-100; b=(rows[pr[k+2]][0]-t0)/1e3+50
qs={}
for s,e,q,kn in rows:
    if a<(s-t0)/1e3<b:
        qs.setdefault(q,len(qs)+1)
        print("%9.1f %9.1f %7.1f q%d %s"%((s-t0)/1e3-a,(e-t0)/1e3-a,(e-s)/1e3,qs[q],kn))
