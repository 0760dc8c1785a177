"""Per-kernel LDS bank-conflict share and wait share from one rocprofv3 --pmc pass (counter_collection CSV):
   python3 tools/pmc_lds.py <counter_collection.csv>"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("swm::", "")
    k = k.split("<lambda")[0][:52]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    rows.append((wc, k, d))
tot = sum(r[0] for r in rows) or 1
print("%-52s %7s %9s %9s %8s" % ("kernel", "wave-cy%", "conflict%", "lds-busy%", "wait%"))
for wc, k, d in sorted(rows, reverse=True)[:22]:
    act = d.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-52s %7.1f %9.1f %9.1f %8.1f" % (k, 100 * wc / tot, 100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / act if act else 0,
                                           100 * act / (d.get("SQ_BUSY_CYCLES", 0) or 1), 100 * d.get("SQ_WAIT_ANY", 0) / wc if wc else 0))
