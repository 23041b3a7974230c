"""One training step out of a rocprofv3 kernel trace: per-stream busy time, the main stream's gaps (with the
kernels around them) and the per-kernel totals of that step. Steps are delimited by k_momentum_update.
Usage: python tools/step_timeline.py <trace.db> [step_index_from_end=1]"""
import collections
import sqlite3
import subprocess
import sys

con = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = con.execute("select d.start, d.end, d.stream_id, s.kernel_name from %s d join %s s on d.kernel_id = s.id "
                   "order by d.start" % (disp, sym)).fetchall()
names = sorted({r[3] for r in rows})
dm = subprocess.run(["/usr/bin/c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names),
                    capture_output=True, text=True).stdout.split("\n")
short = {}
for n, d in zip(names, dm):
    d = d.replace("mtlssl::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    short[n] = d.split("(")[0][:60]
ends = [i for i, r in enumerate(rows) if "k_momentum_update" in r[3]]
spans = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1) if ends[i + 1] - ends[i] > 300]   # real steps
lo, hi = spans[-back]
step = rows[lo:hi]
t0, t1 = step[0][0], step[-1][1]
print("step of %.2f ms, %d dispatches" % ((t1 - t0) / 1e6, len(step)))
by = collections.defaultdict(list)
for s, e, st, n in step:
    by[st].append((s, e, n))
main = max(by, key=lambda k: len(by[k]))
for st, v in by.items():
    print("stream %s: %d kernels, busy %.2f ms, first at +%.2f ms, last end +%.2f ms" %
          (st, len(v), sum(e - s for s, e, _ in v) / 1e6, (v[0][0] - t0) / 1e6, (v[-1][1] - t0) / 1e6))
# device-level: time with no kernel running
ev = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
idle, run, last = 0, 0, t0
for t, d in ev:
    if run == 0:
        idle += t - last
    run += d
    last = t
print("device idle (no kernel on any stream): %.2f ms" % (idle / 1e6))
print("\nmain-stream gaps >= 15 us:")
v = by[main]
for i in range(len(v) - 1):
    g = v[i + 1][0] - v[i][1]
    if g >= 15000:
        print("  +%.2f ms gap %.0f us after %s before %s" % ((v[i][1] - t0) / 1e6, g / 1e3, short[v[i][2]], short[v[i + 1][2]]))
print("\nper-kernel totals on the MAIN stream:")
totm, cntm = collections.Counter(), collections.Counter()
for s_, e_, n_ in by[main]:
    totm[short[n_]] += e_ - s_
    cntm[short[n_]] += 1
for n_, t_ in totm.most_common(30):
    print("  %-62s %4d  %8.3f ms" % (n_, cntm[n_], t_ / 1e6))
for st in by:
    if st == main:
        continue
    print("\nstream %s by kernel:" % st)
    tt, cc = collections.Counter(), collections.Counter()
    for s_, e_, n_ in by[st]:
        tt[short[n_]] += e_ - s_
        cc[short[n_]] += 1
    for n_, t_ in tt.most_common(12):
        print("  %-62s %4d  %8.3f ms" % (n_, cc[n_], t_ / 1e6))
print("\nmain stream, 2-ms bins (busy ms | dominant kernel):")
bins = collections.defaultdict(collections.Counter)
for s_, e_, n_ in by[main]:
    bins[int((s_ - t0) / 2e6)][short[n_]] += e_ - s_
for b in sorted(bins):
    top = bins[b].most_common(2)
    print("  %5.1f ms: %.2f | %s" % (b * 2.0, sum(bins[b].values()) / 1e6, ", ".join("%s %.2f" % (k[:38], v / 1e6) for k, v in top)))
print("\nper-kernel totals of the step (all streams):")
tot = collections.Counter()
cnt = collections.Counter()
for s, e, st, n in step:
    tot[short[n]] += e - s
    cnt[short[n]] += 1
for n, t in tot.most_common(45):
    print("  %-62s %4d  %8.3f ms" % (n, cnt[n], t / 1e6))
print("  total kernel time %.2f ms" % (sum(tot.values()) / 1e6))
