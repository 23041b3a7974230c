"""Per-kernel HBM counters of tools/pmc_hbm.sh: average FETCH_SIZE (x2: gfx950 tallies wide 16-B/lane loads at half,
MI355X_MICROARCH.md "HBM") and WRITE_SIZE per launch of the LAST launches of each kernel (the stand-alone timing loops
of tools/hbm_kernels.py, not the training step before them). Writes <dir>/hbm_kernels_pmc.json."""
import collections
import csv
import glob
import json
import os
import sys

WANT = ["k_roi_crop_pool_fwd", "k_roi_crop_pool_bwd_lds", "k_absmax_bits", "k_rpn_decode_score", "k_rank_partial", "k_rank_scatter", "k_nms_greedy", "k_nms_prune", "k_nms_mask", "k_nms_scan",
        "k_emit_proposals", "k_var_sumsq", "k_momentum_update"]
TAIL = {"k_var_sumsq": 5, "k_momentum_update": 5}     # launches of the stand-alone loop to average (default 5)


def load(root, sub, counter):
    path = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return per


def pick(per, frag):
    for k, v in per.items():
        if frag + "(" in k or k.endswith(frag) or ("::" + frag) in k:
            return v
    return []


root = sys.argv[1]
fe, wr = load(root, "fetch", "FETCH_SIZE"), load(root, "write", "WRITE_SIZE")
hit, miss = load(root, "l2", "TCC_HIT_sum"), load(root, "l2", "TCC_MISS_sum")
out = {}
print("| kernel | launches seen | HBM read / launch (FETCH_SIZE x2) | HBM write / launch (WRITE_SIZE) | L2 hit |\n|---|---|---|---|---|")
for frag in WANT:
    f, w = pick(fe, frag), pick(wr, frag)
    if not f:
        continue
    n = TAIL.get(frag, 5)
    fb = 2.0 * 1024.0 * sum(f[-n:]) / len(f[-n:])
    wb = 1024.0 * sum(w[-n:]) / max(len(w[-n:]), 1)
    h, m = sum(pick(hit, frag)[-n:]), sum(pick(miss, frag)[-n:])
    out[frag] = {"fetch_bytes": fb, "write_bytes": wb, "l2_hit": h / max(h + m, 1.0), "launches_seen": len(f)}
    print("| `%s` | %d | %.2f MB | %.2f MB | %.0f %% |" % (frag, len(f), fb / 1e6, wb / 1e6, 100 * h / max(h + m, 1.0)))
json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate passes over "
                     "tools/hbm_kernels.py (tools/pmc_hbm.sh); FETCH_SIZE x 1 KiB x 2 (gfx950 wide-load correction), "
                     "WRITE_SIZE x 1 KiB uncalibrated; averages over the last launches (the stand-alone loops)",
           "kernels": out}, open(os.path.join(root, "hbm_kernels_pmc.json"), "w"), indent=1)
