import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r["Kernel_Name"].split("(")[0]
    agg[k][0]+=1; agg[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
for k,(n,us) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print(f"{k[:70]:70s} n={n:6d} total={us/1e3:8.2f} ms  avg={us/n:7.1f} us  {100*us/tot:5.1f}%")
