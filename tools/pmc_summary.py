"""Summarise the PMC passes of tools/pmc_bench.sh per kernel: MFMA utilisation, wave-cycle split,
HBM traffic per launch (FETCH_SIZE doubled for wide loads per MI355X_MICROARCH.md's gfx950 note;
WRITE_SIZE as reported), L2 hit rate. Usage: python tools/pmc_summary.py gpurun_out/pmc_bench [top_n] [json_out]"""
import collections
import csv
import json
import os
import sys


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return per


def avg(d, k, name):
    v = d.get(k, {}).get(name, [])
    return sum(v) / len(v) if v else 0.0


def main(root, top=14):
    sq = load(os.path.join(root, "sq", "sq_counter_collection.csv"))
    fe = load(os.path.join(root, "fetch", "fetch_counter_collection.csv"))
    wr = load(os.path.join(root, "write", "write_counter_collection.csv"))
    l2 = load(os.path.join(root, "l2", "l2_counter_collection.csv"))
    rows = []
    for k, c in sq.items():
        g = sum(c["GRBM_GUI_ACTIVE"])
        if not g:
            continue
        wave = sum(c.get("SQ_WAVE_CYCLES", [0])) or 1.0
        hit, miss = sum(l2.get(k, {}).get("TCC_HIT_sum", [0])), sum(l2.get(k, {}).get("TCC_MISS_sum", [0]))
        rows.append(dict(
            kernel=k, launches=len(c["GRBM_GUI_ACTIVE"]), gui_active=g,
            # 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            mfma_util=sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / (g * 128.0),
            wait_any=sum(c.get("SQ_WAIT_ANY", [0])) / wave, wait_inst=sum(c.get("SQ_WAIT_INST_ANY", [0])) / wave,
            active=sum(c.get("SQ_ACTIVE_INST_ANY", [0])) / wave,
            fetch_bytes=2.0 * 1024.0 * avg(fe, k, "FETCH_SIZE"), write_bytes=1024.0 * avg(wr, k, "WRITE_SIZE"),
            l2_hit=hit / max(hit + miss, 1.0)))
    rows.sort(key=lambda r: -r["gui_active"])
    tot = sum(r["gui_active"] for r in rows)
    print("| kernel | launches | share of GPU-active cycles | MFMA util | wave cycles: wait / issue-stall / active | "
          "HBM read per launch (FETCH_SIZE x2) | HBM write per launch | L2 hit |\n|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        print("| `%s` | %d | %.1f %% | %.1f %% | %.0f / %.0f / %.0f %% | %.1f MB | %.1f MB | %.0f %% |" % (
            r["kernel"][:90], r["launches"], 100 * r["gui_active"] / tot, 100 * r["mfma_util"], 100 * r["wait_any"],
            100 * r["wait_inst"], 100 * r["active"], r["fetch_bytes"] / 1e6, r["write_bytes"] / 1e6, 100 * r["l2_hit"]))
    return rows


def kernel_source_sha16():
    """Digest of the source of the roofline kernel (the tile engine, conv_mfma.h): stored with the counters so that
    bench.py can tell whether the committed traffic figure was taken from the kernel it is timing."""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtl_ssl_amd", "csrc")
    h = hashlib.sha256()
    for name in ("conv_mfma.h",):
        h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    rows = main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
    if len(sys.argv) > 3:
        dom = next(r for r in rows if "k_conv_mfma_pw<128, 128, 0" in r["kernel"])
        json.dump({"kernel": dom["kernel"], "launches": dom["launches"],
                   "hbm_read_bytes_per_launch": dom["fetch_bytes"], "hbm_write_bytes_per_launch": dom["write_bytes"],
                   "mfma_util": dom["mfma_util"], "l2_hit": dom["l2_hit"], "kernel_source_sha16": kernel_source_sha16(),
                   "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py "
                             "(tools/pmc_bench.sh); FETCH_SIZE doubled (gfx950 wide-load correction)"},
                  open(sys.argv[3], "w"), indent=1)
