"""Build libmtlssl_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m mtl_ssl_amd.build [--force]

Per-file flags: detection.hip is built with -ffp-contract=off so its float arithmetic is
evaluated in the same order as the CPU oracle (bit-exact integer outputs).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmtlssl_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
SOURCES = {
    "detection.hip": ["-ffp-contract=off"],
    "ops.hip": ["-ffp-contract=off"],   # ROI max-pool arg-max must not flip on fma rounding
    "conv.hip": [],
    "glue.hip": ["-ffp-contract=off"],
    "depthwise.hip": [], "pool_concat.hip": [], "winograd.hip": [],
    "comm.hip": [],                     # RCCL wrappers (host code only; RCCL itself is bound with dlopen)
}


def _digest(path, flags):
    h = hashlib.sha256()
    for f in (path, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_mfma.h"), os.path.join(CSRC, "conv_split.h"), os.path.join(CSRC, "portable_math.h"),
              os.path.join(HERE, "..", "include", "mtlssl_hip.h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src + ".o")
        stamp = obj + ".sha"
        dig = _digest(path, COMMON + extra)
        if (not force and os.path.exists(obj) and os.path.exists(stamp)
                and open(stamp).read() == dig):
            continue
        jobs.append((path, obj, stamp, dig, extra))

    def run(job):
        path, obj, stamp, dig, extra = job
        cmd = [HIPCC] + COMMON + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(dig)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT):
        objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
