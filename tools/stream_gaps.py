"""Per-stream occupancy of a rocprofv3 kernel trace: busy time, idle gaps between consecutive kernels of the
same stream, and the gaps of the whole device. Usage: python tools/stream_gaps.py <results.db> [skip_fraction]"""
import collections
import sqlite3
import sys

import numpy as np

con = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
rows = con.execute("select start, end, stream_id, queue_id from %s order by start" % disp).fetchall()
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print("window %.1f ms, %d dispatches" % ((t1 - t0) / 1e6, len(rows)))
by = collections.defaultdict(list)
for s, e, st, q in rows:
    by[(st, q)].append((s, e))
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e in v)
    gaps = np.array([max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1)]) / 1e3
    small = gaps[gaps < 50]
    print("stream/queue %s: %d kernels, busy %.1f ms, gaps<50us: n=%d sum %.2f ms median %.1f us; gaps>=50us: n=%d sum %.1f ms"
          % (k, len(v), busy / 1e6, len(small), small.sum() / 1e3, np.median(small) if len(small) else 0,
             (gaps >= 50).sum(), gaps[gaps >= 50].sum() / 1e3))
