"""GPU idle gaps inside the last proof of a rocprofv3 kernel trace: union of kernel intervals vs span, the largest gaps
and what ran before / after each."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
nproof = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rows = list(db.execute("select name,start,end from kernels order by start"))
br = [i for i, r in enumerate(rows) if 'msm_bucket_reduce' in r[0]]
idx = [i for i, r in enumerate(rows) if r[0].startswith('swm::msm_digits')]
start = idx[-nproof] - 60
sub = rows[start:br[-1] + 2]
ev = sorted((r[1], r[2], r[0]) for r in sub)
gaps = []
cs, ce, last = ev[0][0], ev[0][1], ev[0][2]
busy = 0
for s, e, n in ev[1:]:
    if s > ce:
        gaps.append((s - ce, (ce - ev[0][0]) / 1e3, last.split('(')[0][:40], n.split('(')[0][:40]))
        busy += ce - cs
        cs, ce, last = s, e, n
    elif e > ce:
        ce, last = e, n
busy += ce - cs
span = ev[-1][1] - ev[0][0]
print("span %.2f ms  busy %.2f ms  idle %.2f ms in %d gaps" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps)))
for g in sorted(gaps, reverse=True)[:14]:
    print("  gap %7.1f us at %8.1f us  after %-40s before %s" % (g[0] / 1e3, g[1], g[2], g[3]))
