"""What the main stream of a training step does besides convolutions, from a rocprofv3 kernel trace of bench.py
(steps are delimited by k_momentum_update): conv-family kernel time, the top non-conv kernels, and the time the main
stream spends waiting (no kernel of its own running: events of the side streams, launch gaps).
Usage: python tools/non_conv_breakdown.py <trace.db> [top_n=10]"""
import collections
import sqlite3
import subprocess
import sys

CONV = ("k_conv_", "k_wino_", "k_splitk_", "k_wgrad_reduce", "k_colsum", "k_s2d", "k_parity", "k_pad_rows", "k_small_reduce",
        "k_gemm_small", "k_split_", "k_stem")


def main(path, top=10):
    con = sqlite3.connect(path)
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
        short[n] = d.split("(")[0][:70]
    ends = [i for i, r in enumerate(rows) if "k_momentum_update" in r[3]]
    spans = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1) if ends[i + 1] - ends[i] > 100]
    if not spans:
        print("no complete step in the trace")
        return
    spans = spans[len(spans) // 2:]                      # the later steps (plans settled)
    nsteps = len(spans)
    tot = collections.Counter()
    cnt = collections.Counter()
    step_ms = conv_ms = busy_ms = 0.0
    side_busy = collections.Counter()
    for lo, hi in spans:
        step = rows[lo:hi]
        main_stream = step[-1][2]                        # the optimizer runs on the step's main stream
        step_ms += (step[-1][1] - step[0][0]) / 1e6
        for s, e, st, n in step:
            if st != main_stream:
                side_busy[st] += (e - s) / 1e6
                continue
            busy_ms += (e - s) / 1e6
            nm = short[n]
            if nm.startswith(CONV):
                conv_ms += (e - s) / 1e6
            else:
                tot[nm] += (e - s) / 1e3
                cnt[nm] += 1
    non_conv = sum(tot.values()) / 1e3
    print("%d steps of %.2f ms; main stream per step: busy %.2f ms = conv family %.2f ms + other kernels %.2f ms; waiting "
          "(no kernel of its own: side-stream events, launch gaps) %.2f ms" % (
              nsteps, step_ms / nsteps, busy_ms / nsteps, conv_ms / nsteps, non_conv / nsteps, (step_ms - busy_ms) / nsteps))
    print("side streams busy per step: %s" % ", ".join("%.2f ms" % (v / nsteps) for v in sorted(side_busy.values(), reverse=True)))
    print("\n| non-conv kernel on the main stream | launches / step | us / step | avg us |\n|---|---|---|---|")
    for nm, us in tot.most_common(top):
        print("| `%s` | %.1f | %.1f | %.1f |" % (nm, cnt[nm] / nsteps, us / nsteps, us / cnt[nm]))
    rest = sum(us for nm, us in tot.most_common()[top:])
    print("| (all others) | %.1f | %.1f | |" % (sum(c for nm, c in cnt.items() if nm not in dict(tot.most_common(top))) / nsteps, rest / nsteps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
