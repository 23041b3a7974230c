"""Achieved HBM rate of the Winograd transform kernels from a rocprofv3 kernel trace of the default bench:
groups the dispatches of k_wino_input / k_wino_output / k_wino_dy by grid size (= layer shape) and prices the
big second-stage layers (C = K = 512, 7x7 maps, M7: 121 planes) at their algorithmic bytes.
Usage: python tools/wino_transform_rates.py <results.db>"""
import sqlite3
import subprocess
import sys  # noqa: I001

con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = con.execute("select s.kernel_name, d.grid_size_x, count(*), avg(d.end - d.start) from %s d join %s s on "
                   "d.kernel_id = s.id where s.kernel_name like '%%k_wino_%%' and s.kernel_name not like '%%gemm%%' "
                   "group by s.kernel_name, d.grid_size_x order by 3 * 4 desc" % (disp, sym)).fetchall()
names = subprocess.run(["c++filt"], input="\n".join(r[0].replace(".kd", "") for r in rows), capture_output=True,
                       text=True).stdout.split("\n")
print("| kernel | threads | calls | avg us | GB/s if C = 512 on 7x7 maps (M7) |\n|---|---|---|---|---|")
for n, r in zip(names, rows):
    short = n.replace("mtlssl::(anonymous namespace)::", "").replace("void ", "").replace("float __vector(4)", "float4")
    short = short[:short.index(">(") + 1] if ">(" in short else short.split("(")[0]
    threads = r[1]
    rate = ""
    if "M7" in short and ("k_wino_input" in short or "k_wino_output" in short or "k_wino_dy" in short):
        maps = threads / 512.0                      # one thread per (map, channel)
        byt = maps * 512 * 4 * (49 + 121)           # activations once + 121 planes once
        rate = "%.0f" % (byt / (r[3] * 1e-9) / 1e9)
    print("| `%s` | %d | %d | %.1f | %s |" % (short[:70], threads, r[2], r[3] / 1e3, rate))
