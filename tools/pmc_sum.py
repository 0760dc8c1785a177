import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if sys.argv[2] not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(k, r.get("Dispatch_Id"))
    if key not in seen: seen.add(key); n[k]+=1
for k, d in acc.items():
    print(k[:60], "dispatches", n[k])
    for c, v in sorted(d.items()): print("   %-28s %16.0f per dispatch" % (c, v / max(n[k],1)))
