"""Kernel-by-kernel listing of a window of one training step out of a rocprofv3 kernel trace (steps are delimited by
k_momentum_update): start offset, duration, gap to the previous kernel on the same stream, stream, grid, kernel.
Usage: python tools/kernel_sequence.py <trace.db> [t0_ms=0] [t1_ms=5] [step_index_from_end=1]"""
import sqlite3
import subprocess
import sys

con = sqlite3.connect(sys.argv[1])
w0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
w1 = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
back = int(sys.argv[4]) if len(sys.argv) > 4 else 1
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in con.execute("pragma table_info(%s)" % disp)]
gx = "d.grid_size_x" if "grid_size_x" in cols else "0"
wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else "1"
rows = con.execute("select d.start, d.end, d.stream_id, s.kernel_name, %s, %s from %s d join %s s on d.kernel_id = s.id "
                   "order by d.start" % (gx, wx, disp, sym)).fetchall()
names = sorted({r[3] for r in rows})
dm = subprocess.run(["/usr/bin/c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names),
                    capture_output=True, text=True).stdout.split("\n")
short = {}
for n, d in zip(names, dm):
    d = d.replace("mtlssl::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    short[n] = d.split("(")[0][:56]
ends = [i for i, r in enumerate(rows) if "k_momentum_update" in r[3]]
spans = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1) if ends[i + 1] - ends[i] > 300]
lo, hi = spans[-back]
step = rows[lo:hi]
t0 = step[0][0]
last = {}
streams = {}
print("%9s %8s %7s  st %8s  kernel" % ("start_ms", "dur_us", "gap_us", "blocks"))
for s, e, st, n, g, w in step:
    sid = streams.setdefault(st, len(streams))
    gap = (s - last[st]) / 1e3 if st in last else 0.0
    last[st] = e
    off = (s - t0) / 1e6
    if w0 <= off < w1:
        print("%9.3f %8.1f %7.1f  %2d %8d  %s" % (off, (e - s) / 1e3, gap, sid, (g // max(w, 1)) if g else 0, short[n]))
