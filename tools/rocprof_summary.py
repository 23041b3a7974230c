"""Summarise a rocprofv3 `--kernel-trace` results.db as a markdown table (per-kernel calls, total,
average, share). Usage: python tools/rocprof_summary.py <results.db> [top_n]"""
import os
import sqlite3
import subprocess
import sys

CXXFILT = "/usr/bin/c++filt"


def demangle(names):
    if not os.path.exists(CXXFILT):
        return names
    clean = [n[:-3] if n.endswith(".kd") else n for n in names]
    out = subprocess.run([CXXFILT], input="\n".join(clean), capture_output=True, text=True).stdout.split("\n")
    return out[:len(names)]


def main(path, top=30):
    con = sqlite3.connect(path)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = con.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start) from %s d join %s s "
        "on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (disp, sym)).fetchall()
    dm = demangle([r[0] for r in rows])
    rows = [(d,) + tuple(r[1:]) for d, r in zip(dm, rows)]
    total = sum(r[2] for r in rows)
    print("Total kernel time %.1f ms over %d dispatches, %d distinct kernels\n" %
          (total / 1e6, sum(r[1] for r in rows), len(rows)))
    print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
    for name, n, tot, avg in rows[:top]:
        print("| `%s` | %d | %.0f | %.1f | %.2f |" % (name[:110], n, tot / 1e3, avg / 1e3, 100.0 * tot / total))
    rest = rows[top:]
    if rest:
        print("| (%d more kernels) | %d | %.0f | | %.2f |" %
              (len(rest), sum(r[1] for r in rest), sum(r[2] for r in rest) / 1e3,
               100.0 * sum(r[2] for r in rest) / total))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
