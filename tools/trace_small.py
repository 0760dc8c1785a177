"""Summarise the last proof of a rocprofv3 kernel trace (results.db): span, busy time and per-kernel totals."""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
nm_per = int(sys.argv[2]) if len(sys.argv) > 2 else 17
rows = list(db.execute("select name,start,end,stream_id,grid_x,workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if r[0].startswith('swm::msm_digits')]
sub = rows[idx[-nm_per] - 40:]
t0 = sub[0][1]
dur = collections.defaultdict(float); n = collections.Counter(); mx = collections.defaultdict(float)
for r in sub:
    k = r[0].split('(')[0][:44]
    d = (r[2] - r[1]) / 1e3
    dur[k] += d; n[k] += 1; mx[k] = max(mx[k], d)
ev = sorted((r[1], r[2]) for r in sub)
busy = 0; cs, ce = ev[0]
for s, e in ev[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("span %.0f us  busy %.0f us  kernels %d" % ((sub[-1][2] - t0) / 1e3, busy / 1e3, len(sub)))
for k, v in sorted(dur.items(), key=lambda x: -x[1])[:16]:
    print("%8.1f us %4d  max %7.1f  %s" % (v, n[k], mx[k], k))
