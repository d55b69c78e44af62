import csv,glob,sys
out=sys.argv[1]
f=glob.glob(out+"/**/kt_kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:110]) for r in csv.DictReader(open(f))]
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2].startswith("preprocess_kernel")]
a,b=idx[-5],idx[-4]
prev=None
for s,e,n in rows[a-3:b+1]:
    print("%-112s dur %7.1f gap %6.1f"%(n,(e-s)/1e3,((s-prev)/1e3 if prev else 0))); prev=e
