"""Host-side operator layer over the C ABI (libmtlssl_hip.so).

torch is used only as the device allocator / stream provider: every function takes and
returns torch CUDA tensors whose storage is handed to the HIP kernels as raw pointers.
Names follow the reference operators they replace (see include/mtlssl_hip.h).
"""
import ctypes
import os

import torch

from .lib import ConvDesc, lib, ptr

EPI_BIAS, EPI_RESIDUAL, EPI_RELU, EPI_TANH, EPI_RELU6, EPI_MASK, EPI_ACCUM, EPI_MASK6 = 1, 2, 4, 8, 16, 32, 64, 128
f32 = torch.float32
i32 = torch.int32


# torch.cuda.current_stream() builds a Stream object through five Python frames (~3 us; 340 calls per MobileNet step
# were a third of that configuration's launch-thread time, tools/lab/host_profile.py); the raw handle is one C call
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=f32):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.dtype, t.is_contiguous())
    return t


def _chk_y(d, t):
    """The output-side tensor of a convolution (y / dy): dense, or — d.ldy set — the channel slice [..., c0:c0+K] of a
    contiguous NHWC map with d.ldy channels (a torch view: its data_ptr() is the slice's first element)."""
    if not d.ldy:
        return _chk(t)
    assert t.is_cuda and t.dtype == f32 and t.dim() == 4 and t.shape[-1] == d.K and t.stride(-1) == 1 \
        and t.stride(-2) == d.ldy and t.stride(-3) == d.ldy * t.shape[-2] and t.data_ptr() % 16 == 0, \
        (tuple(t.shape), t.stride(), d.ldy)
    return t


_ws_cache = {}


POISON_WS = os.environ.get("MTLSSL_POISON_WS") == "1"


def workspace(nbytes, key="default", device=None):
    """Grow-only device scratch buffers, one per key (caller-owned workspaces of the C ABI).
    MTLSSL_POISON_WS=1 (the test suite sets it) fills a buffer with NaN bit patterns every time it is handed out, so
    that no entry point can get by on what an earlier call left in its workspace."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    # one buffer per (purpose, device, stream): kernels on different streams may run concurrently
    k = (key, device.index, _stream())
    buf = _ws_cache.get(k)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[k] = buf
    if POISON_WS:
        buf.fill_(255)
    return buf


# ------------------------------------------------------------------------------- conv family
def same_pad(in_size, k, stride, dilation=1):
    """TF 'SAME': (pad_before, out_size)."""
    k_eff = (k - 1) * dilation + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return total // 2, out


def conv_desc(x_shape, w_shape, stride=1, dilation=1, padding="SAME", ldy=0):
    """padding: 'SAME' | 'VALID' | 'RESNET_SAME' (slim/nets/resnet_utils.py:77-122 conv2d_same).
    ldy: row stride (floats) of the output-side tensor — y of the forward, dy of dgrad / wgrad — when it is a channel
    slice of a wider NHWC map (tf.concat(axis=3) folded into its producers / the consumers of its gradient); 0 = dense."""
    N, H, W, C = x_shape
    R, S, C2, K = w_shape
    assert C == C2, (x_shape, w_shape)
    if padding == "RESNET_SAME" and stride > 1:
        k_eff = R + (R - 1) * (dilation - 1)
        pt = pl = (k_eff - 1) // 2
        OH = (H + k_eff - 1 - k_eff) // stride + 1
        OW = (W + k_eff - 1 - k_eff) // stride + 1
    elif padding in ("SAME", "RESNET_SAME"):
        pt, OH = same_pad(H, R, stride, dilation)
        pl, OW = same_pad(W, S, stride, dilation)
    elif padding == "VALID":
        pt = pl = 0
        OH = (H - ((R - 1) * dilation + 1)) // stride + 1
        OW = (W - ((S - 1) * dilation + 1)) // stride + 1
    else:
        raise ValueError(padding)
    return ConvDesc(N, H, W, C, K, R, S, OH, OW, stride, dilation, pt, pl, int(ldy) if int(ldy) != K else 0)


def desc_is_pointwise(d):
    """The layer runs on the tile engine's pointwise instantiation (k_conv_mfma_pw / k_conv_glds_pw): 1x1, stride 1,
    no padding, output map = input map (conv_is_pointwise in csrc/conv_mfma.h)."""
    return (d.R == 1 and d.S == 1 and d.stride == 1 and d.dilation == 1 and d.pad_t == 0 and d.pad_l == 0
            and d.OH == d.H and d.OW == d.W)


class ConvProfiler:
    """Per-launch HIP-event timing of the conv family, keyed by (mode, tile config, pointwise). Events are
    recorded on the stream the kernels are launched on (torch's current stream). Used by bench.py
    for the `roofline` object; off by default (PROFILER is None)."""
    MODES = ("fwd", "dgrad", "wgrad")

    def __init__(self, only=None):
        """only: optional (mode, tile config, pointwise) — time just that kernel (two events per launch cost
        ~1 us of stream time each; timing every conv of the step costs ~2 % of the step)."""
        self.pending = []          # (key, flops, start_event, end_event)
        self.calls = []            # (launch class, stream handle, executed MFMA flops, start_event, end_event)
        self.only = only

    def begin(self, d=None, mode=None):
        if self.only is not None and (mode != self.only[0]
                                      or lib().conv2d_tile_config(ctypes.byref(d), mode) != self.only[1]
                                      or desc_is_pointwise(d) != self.only[2]):
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, d, mode, start):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        cfg = lib().conv2d_tile_config(ctypes.byref(d), mode)
        flops = 2.0 * d.N * d.OH * d.OW * d.K * d.C * d.R * d.S
        self.pending.append(((self.MODES[mode], cfg, desc_is_pointwise(d)), flops, start, e,
                             lib().conv2d_num_dispatches(ctypes.byref(d), mode)))
        mf = int(lib().conv2d_executed_macs(ctypes.byref(d), mode, 1))
        self.calls.append((conv_class(cfg, mf), torch.cuda.current_stream().cuda_stream, 2.0 * mf, start, e))

    def summary(self):
        """{(mode, cfg, pointwise): dict(launches, dispatches, seconds, flops)} — call after
        torch.cuda.synchronize(). launches = calls; dispatches = launches of the MFMA kernel."""
        out = {}
        for key, flops, s, e, nd in self.pending:
            r = out.setdefault(key, {"launches": 0, "dispatches": 0, "seconds": 0.0, "flops": 0.0})
            r["launches"] += 1
            r["dispatches"] += nd
            r["seconds"] += s.elapsed_time(e) * 1e-3
            r["flops"] += flops
        return out


    def class_summary(self, main_stream):
        """{class: {"main_ms", "side_ms", "calls", "executed_tflop", "tflops_on_main"}} — per-call wall time on the
        stream the call was issued to (a Winograd call = its transforms + its GEMM stack), split by whether that
        stream is the step's main stream. Call after torch.cuda.synchronize()."""
        out = {}
        for cls, st, fl, s, e in self.calls:
            r = out.setdefault(cls, {"main_ms": 0.0, "side_ms": 0.0, "calls": 0, "executed_tflop": 0.0, "_fm": 0.0})
            ms = s.elapsed_time(e)
            r["main_ms" if st == main_stream else "side_ms"] += ms
            r["calls"] += 1
            r["executed_tflop"] += fl / 1e12
            if st == main_stream:
                r["_fm"] += fl
        for r in out.values():
            fm = r.pop("_fm")
            r["tflops_on_main"] = (fm / (r["main_ms"] * 1e-3) / 1e12) if r["main_ms"] > 0 else None
        return out


PROFILER = None
PHASE_HOOK = None      # tools/phase_times.py: callable(name) invoked at the step's phase boundaries


class HbmProfiler:
    """Live HIP-event timing of the HBM-bound kernel families inside a timed region (bench.py: `roofline` of a
    configuration whose dominant kernel is HBM-bound, and the in-step rows of `hbm_kernels`): per family the compulsory
    bytes (inputs once + outputs once) and the launch durations on the stream they were issued to."""

    def __init__(self, families=("depthwise_fwd", "depthwise_dgrad", "depthwise_wgrad", "psroi_fwd", "psroi_bwd",
                                 "roi_crop_pool_fwd")):
        self.families = set(families)
        self.rows = []           # (family, bytes, start_event, end_event)

    def begin(self, family):
        if family not in self.families:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, family, nbytes, start):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.rows.append((family, int(nbytes), start, e))

    def summary(self):
        """{family: {"launches", "seconds", "bytes"}} — after torch.cuda.synchronize()."""
        out = {}
        for fam, nb, s, e in self.rows:
            r = out.setdefault(fam, {"launches": 0, "seconds": 0.0, "bytes": 0})
            r["launches"] += 1
            r["seconds"] += s.elapsed_time(e) * 1e-3
            r["bytes"] += nb
        return out


HBM_PROFILER = None


def _hbm_begin(family):
    return HBM_PROFILER.begin(family) if HBM_PROFILER is not None else None


def conv_class(cfg, mfma_macs):
    """Launch class of a conv-family call from its plan code (mtlssl_conv2d_tile_config)."""
    if cfg < 0:
        return "mfma_padded_or_space_to_depth" if mfma_macs else "valu_fallback"
    alg, shape = plan_code_algorithm(cfg), cfg % 4
    if alg == 2:
        return "winograd_M7"
    if alg == 1:
        return "winograd_F43"
    return "direct_128x128_256x128" if shape in (0, 3) else "direct_small_tiles"


ENGINE1_CFG0 = 12      # plan codes 12..23: the same algorithms / tile shapes as 0..11 on the LDS-DMA tile engine
PLAN_CODES = tuple(range(24))


def plan_code_algorithm(cfg):
    """0: direct implicit GEMM, 1: Winograd F(4x4,3x3), 2: whole-7-span Winograd (either tile engine)."""
    return (cfg % ENGINE1_CFG0) // 4


def plan_code_engine(cfg):
    """0: operands staged through registers (k_conv_mfma / k_wino_gemm), 1: LDS-DMA (k_conv_glds / k_wino_glds)."""
    return 1 if cfg >= ENGINE1_CFG0 else 0


class FlopAccount:
    """Executed multiply-accumulates of the conv family per launch class, summed from the library's own plan
    registry (mtlssl_conv2d_executed_macs: direct = M*N*K of the implicit GEMM, Winograd = the transformed-domain
    GEMM stack, ...). Pure host arithmetic, memoised per problem: cheap enough to leave on during a timed region.
    bench.py divides the sums by the step time for `whole_step.executed_tflops`."""

    def __init__(self):
        self.rows = {}            # class -> [calls, mfma_macs, valu_macs, direct_algorithm_macs]
        self._memo = {}

    def add(self, d, mode, times=1):
        key = (d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.OH, d.OW, d.stride, d.dilation, d.pad_t, d.pad_l, mode)
        rec = self._memo.get(key)
        if rec is None:
            mf = int(lib().conv2d_executed_macs(ctypes.byref(d), mode, 1))
            va = int(lib().conv2d_executed_macs(ctypes.byref(d), mode, 0))
            cfg = lib().conv2d_tile_config(ctypes.byref(d), mode)
            direct = d.N * d.OH * d.OW * d.K * d.C * d.R * d.S
            rec = self._memo[key] = (conv_class(cfg, mf), mf, va, direct)
        r = self.rows.setdefault(rec[0], [0, 0, 0, 0])
        r[0] += times
        r[1] += rec[1] * times
        r[2] += rec[2] * times
        r[3] += rec[3] * times


ACCOUNT = None         # a FlopAccount while bench.py counts executed FLOPs


class JoinTimer:
    """HIP-event pairs around the points where the step's MAIN stream waits for a side stream (bench.py's
    whole_step.main_stream_idle_ms): `wait_on` records one event just before and one just after the wait on the waiting
    stream, so their distance is the time that stream had nothing of its own to run at that join — measured un-profiled,
    a handful of event pairs per step. Off by default (JOIN_TIMER is None)."""

    def __init__(self):
        self.pairs = []

    def summary(self, steps):
        """{"total_ms_per_step", "by_join": {tag: ms_per_step}} — after torch.cuda.synchronize()."""
        by = {}
        for tag, a, b in self.pairs:
            by[tag] = by.get(tag, 0.0) + a.elapsed_time(b)
        steps = max(int(steps), 1)
        return {"total_ms_per_step": round(sum(by.values()) / steps, 3),
                "by_join": {k: round(v / steps, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}}


JOIN_TIMER = None


def wait_on(other, tag, cur=None):
    """The current (or given) stream waits for everything enqueued on stream `other` (or for event `other`)."""
    cur = cur if cur is not None else torch.cuda.current_stream()
    jt = JOIN_TIMER
    if jt is not None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
    if isinstance(other, torch.cuda.Event):
        cur.wait_event(other)
    else:
        cur.wait_stream(other)
    if jt is not None:
        b.record(cur)
        jt.pairs.append((tag, a, b))


def mark(name):
    if PHASE_HOOK is not None:
        PHASE_HOOK(name)

# ---- tile autotuning. The time model mis-ranks tiles on some small GEMMs, so tile choices are
# measured: `conv_plans.json` (next to this file, generated on an MI355X by tools/tune_plans.py) pins
# the winner for every layer shape of the shipped configurations — no trial launches at run time and
# the same plan on every run and rank; a (problem, mode) that is not in the table is timed on first
# use on the caller's tensors (scratch outputs). Either way the choice is handed to the library's
# plan registry (mtlssl_conv2d_force_config) only when it beat the planner's own by > 5 %.
# The on-line tuner is a measurement tool (tools/tune_plans.py, MTLSSL_AUTOTUNE=1): which tile wins a timing race
# differs between machines, and with it the summation order, the ReLU flips and the near-ties of a step. By default a
# problem runs on the committed plan table (conv_plans.json) or, when it is not in the table, on the library's planner
# — the same kernels on every box.
AUTOTUNE = os.environ.get("MTLSSL_AUTOTUNE", "0") != "0"
_AUTOTUNE_DEFAULT = AUTOTUNE
TUNE_RUNS = int(os.environ.get("MTLSSL_TUNE_RUNS", "4"))
TUNE_MARGIN = float(os.environ.get("MTLSSL_TUNE_MARGIN", "0.05"))   # a candidate must beat the planner's choice by this much
TUNE_ENGINES = os.environ.get("MTLSSL_TUNE_ENGINES", "1") != "0"     # 0: the autotuner leaves the LDS-DMA tile engine out
_PLAN_FILE = os.environ.get("MTLSSL_PLAN_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_plans.json")
_tuned = {}
_plan_db = None


def _plans():
    global _plan_db
    if _plan_db is None:
        _plan_db = {}
        if os.path.exists(_PLAN_FILE) and os.environ.get("MTLSSL_PLAN_DB", "1") != "0":
            import json
            _plan_db = json.load(open(_PLAN_FILE))["plans"]
    return _plan_db


def save_plans(path=_PLAN_FILE):
    """Write every measured (problem, mode) -> tile choice (merged over what the file already holds)."""
    import json
    plans = dict(_plans())
    for key, val in _tuned.items():
        if val is not None:
            plans[",".join(str(v) for v in key)] = val[1] if val[1] != val[0] else -1
    json.dump({"comment": "mode,N,H,W,C,K,R,S,OH,OW,stride,dilation,pad_t,pad_l -> forced tile "
                          "(0: 128x128, 1: 128x64, 2: 64x64, 3: 256x128; 4-7: Winograd F(4x4,3x3) with GEMM tile 0-3; 8-11: whole-7-span Winograd with GEMM "
                          "tile 0-3; 12-23: the same twelve on the LDS-DMA tile engine; "
                          "-1: keep the planner's choice); "
                          "measured on MI355X",
               "plans": dict(sorted(plans.items()))}, open(path, "w"), indent=0)


def _plan_key(d, mode):
    return (mode, d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.OH, d.OW, d.stride, d.dilation, d.pad_t, d.pad_l)


WINO_CFG0 = 4          # tile codes WINO_CFG0 + t: Winograd F(4x4,3x3) with GEMM tile t
WINO7_CFG0 = 8         # tile codes WINO7_CFG0 + t: whole-7-span Winograd (maps of 7k x 7k) with GEMM tile t


def force_conv_config(d, mode, cfg):
    """Pin the algorithm / tile of one (problem, mode) — 0..3 direct tiles, 4..7 Winograd F(4x4,3x3),
    8..11 whole-7-span Winograd, -1 back to the planner — and keep the autotuner away from it."""
    _tuned[_plan_key(d, mode)] = None
    lib().conv2d_force_config(ctypes.byref(d), mode, int(cfg))
    return lib().conv2d_tile_config(ctypes.byref(d), mode)


def reset_tuning(use_plan_db=True, autotune=None):
    """Forget every measured / pinned plan: each (problem, mode) seen so far goes back to the library's planner, and
    the next call decides again — from conv_plans.json when `use_plan_db`, by timing when `autotune`. Tests that force
    one algorithm (set_winograd(0 / 2)) call reset_tuning(False, False) first: a tile pinned by an earlier test of
    the same process would otherwise override the mode for that problem."""
    global _plan_db, AUTOTUNE
    import collections
    D = collections.namedtuple("D", "N H W C K R S OH OW stride dilation pad_t pad_l")
    from .lib import ConvDesc
    for key in list(_tuned):
        mode, vals = key[0], key[1:]
        d = ConvDesc(**D(*vals)._asdict())
        lib().conv2d_force_config(ctypes.byref(d), mode, -1)
    _tuned.clear()
    _plan_db = None if use_plan_db else {}
    AUTOTUNE = _AUTOTUNE_DEFAULT if autotune is None else bool(autotune)


def set_winograd(mode):
    """0: direct convolution only, 1: plan registry / time model (default), 2: Winograd F(4x4,3x3) for every
    eligible 3x3 stride-1 layer. Returns the previous mode (mode=-1 only queries)."""
    return lib().conv2d_set_winograd(int(mode))


def set_pointwise(on):
    """Test switch: 0 routes 1x1 stride-1 layers through the tile engine's general gather instantiation (bit-identical,
    slower) instead of the pointwise one. Returns the previous setting (on=-1 only queries)."""
    return lib().conv2d_set_pointwise(int(on))


def set_fp32_engine(mode):
    """0: the native fp32 MFMA engine (default), 1: large implicit GEMMs on the exact split-bf16 engine
    (csrc/conv_split.h; MTLSSL_FP32_ENGINE=split sets the initial value). Returns the previous mode; -1 only queries."""
    return lib().conv2d_set_fp32_engine(int(mode))


def _autotune(d, mode, run):
    key = _plan_key(d, mode)
    if key in _tuned:
        return
    _tuned[key] = None
    L, ref = lib(), ctypes.byref(d)
    default = L.conv2d_tile_config(ref, mode)
    if default < 0:                                   # not on the MFMA path
        return
    known = _plans().get(",".join(str(v) for v in key))
    if known is not None:
        if known >= 0:
            L.conv2d_force_config(ref, mode, int(known))
        return
    if not AUTOTUNE:
        return
    times = {}
    for cfg in (PLAN_CODES if TUNE_ENGINES else PLAN_CODES[:ENGINE1_CFG0]):
        L.conv2d_force_config(ref, mode, cfg)
        if L.conv2d_tile_config(ref, mode) != cfg:    # this tile is not available for the problem
            continue
        run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(TUNE_RUNS):
            run()
        e.record()
        e.synchronize()
        times[cfg] = s.elapsed_time(e)
    L.conv2d_force_config(ref, mode, -1)
    best = min(times, key=times.get)
    if not (best != default and times[best] < (1.0 - TUNE_MARGIN) * times.get(default, float("inf"))):
        best = default
    if best != default:
        L.conv2d_force_config(ref, mode, best)
    _tuned[key] = (default, best, times)


class FilterXfCache:
    """Transformed filters (U = G g G^T) of a model's Winograd layers, kept across the convolution calls of a step
    and refreshed once per optimizer step (`refresh`, from the model's refold) instead of being recomputed by
    every forward / dgrad call — the weights only change in the optimizer. Only layers whose weight tensors
    persist and change nowhere else may use it (nn.ConvBN / nn.Conv pass it; ad-hoc tensors never do).
    Entries are keyed by (weight storage, filter shape, mode, Winograd variant). A refresh may run on a side
    stream: consumers wait for its event the first time they touch an entry on a given stream."""

    def __init__(self):
        self.entries = {}

    def get(self, d, mode, w):
        """-> (U tensor or None, variant) for a forward (mode 0) / dgrad (mode 1) call of `d` with filter `w`."""
        variant = lib().conv2d_filter_xf_variant(ctypes.byref(d), mode)
        if variant < 0:
            return None, -1
        key = (w.data_ptr(), d.C, d.K, mode, variant)
        cur = torch.cuda.current_stream()
        e = self.entries.get(key)
        if e is None:
            nbytes = lib().conv2d_filter_xf_bytes(ctypes.byref(d), mode)
            e = {"U": torch.empty(int(nbytes) // 4, dtype=f32, device=w.device), "d": d, "mode": mode, "variant": variant,
                 "w": w, "event": torch.cuda.Event(), "waited": {cur.cuda_stream}}
            lib().conv2d_transform_filter(ctypes.byref(d), mode, variant, ptr(_chk(w)), ptr(e["U"]), cur.cuda_stream)
            e["event"].record(cur)
            self.entries[key] = e
        elif cur.cuda_stream not in e["waited"]:
            wait_on(e["event"], "filter_cache_refresh", cur)
            e["waited"].add(cur.cuda_stream)
        return e["U"], variant

    # Entries are refreshed in chunks of consecutive entries of one variant (dict order = order of first use in a step:
    # the trunk's forward layers first, then the towers, then the dgrad forms in backward order), at most this many bytes
    # of transformed filters per launch, each chunk with an event of its own: the first Winograd layer of the next forward
    # waits for the first small launch, not for the transform of every filter of the model (1.3 GB at configs[1] — a
    # 0.7-ms stall of the main stream per step, measured with ops.JoinTimer)
    CHUNK_BYTES = int(float(os.environ.get("MTLSSL_XF_REFRESH_CHUNK_MB", "48")) * (1 << 20))

    def _tables(self):
        """Device pointer tables per chunk, rebuilt only when the set of entries changed."""
        if getattr(self, "_tab_n", -1) != len(self.entries):
            chunks, cur, cur_bytes = [], [], 0
            for e in self.entries.values():
                nb = e["U"].numel() * 4
                if cur and (cur[0]["variant"] != e["variant"] or (cur_bytes + nb > self.CHUNK_BYTES > 0)):
                    chunks.append(cur)
                    cur, cur_bytes = [], 0
                cur.append(e)
                cur_bytes += nb
            if cur:
                chunks.append(cur)
            self._tab = []
            for es in chunks:
                dev = es[0]["U"].device
                self._tab.append(dict(
                    variant=es[0]["variant"], entries=es,
                    n=len(es), max_ck=max(e["d"].C * e["d"].K for e in es),
                    w=torch.tensor([e["w"].data_ptr() for e in es], dtype=torch.int64, device=dev),
                    u=torch.tensor([e["U"].data_ptr() for e in es], dtype=torch.int64, device=dev),
                    ck=torch.tensor([e["d"].C * e["d"].K for e in es], dtype=torch.int64, device=dev),
                    flip=torch.tensor([int(e["mode"] == 1) for e in es], dtype=i32, device=dev)))
            self._tab_n = len(self.entries)
        return self._tab

    def refresh(self, stream=None):
        """Recompute every entry from the current weights, one launch per chunk — on `stream` (behind everything
        enqueued on the current stream so far) or on the current stream."""
        if not self.entries:
            return
        cur = torch.cuda.current_stream()
        run_on = stream if stream is not None else cur
        tabs = self._tables()          # (host->device copies of a rebuild happen before the fork below)
        if stream is not None:
            stream.wait_stream(cur)
        for t in tabs:
            lib().conv2d_transform_filters(t["variant"], t["n"], ptr(t["w"]), ptr(t["u"]), ptr(t["ck"]), ptr(t["flip"]),
                                           t["max_ck"], run_on.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(run_on)
            for e in t["entries"]:
                e["event"], e["waited"] = ev, {run_on.cuda_stream}


KEEP_INPUT_XF = os.environ.get("MTLSSL_KEEP_INPUT_XF", "1") != "0"


def _shared_input_variant(d):
    """The Winograd variant both the forward and the filter gradient of `d` are planned with (-1: none in common)."""
    if not (KEEP_INPUT_XF and d.R == 3 and d.S == 3):
        return -1
    v = lib().conv2d_filter_xf_variant(ctypes.byref(d), 0)
    return v if (v >= 0 and v == lib().conv2d_filter_xf_variant(ctypes.byref(d), 2)) else -1


def conv2d_fwd(d, x, w, bias=None, residual=None, epilogue=0, out=None, xf_cache=None, keep_input_xf=None):
    """keep_input_xf: a dict (owned by the layer) that receives {x.data_ptr(): (V, variant)} when the forward
    and the filter gradient of this problem share a Winograd input transform; conv2d_wgrad takes it back."""
    if d.ldy:
        assert out is not None, "a strided forward writes into the caller's concatenated map"
        _chk_y(d, out)
    y = out if out is not None else torch.empty((d.N, d.OH, d.OW, d.K), dtype=f32, device=x.device)
    if PROFILER is None:
        def run():
            nb_ = lib().conv2d_workspace_bytes(ctypes.byref(d), 0)
            ws_ = workspace(nb_, "splitk", x.device) if nb_ else None
            lib().conv2d_fwd(ctypes.byref(d), ptr(_chk(x)), ptr(_chk(w)), ptr(bias), ptr(residual), ptr(y), epilogue,
                             ptr(ws_), _stream())
        _autotune(d, 0, run)
    t0 = PROFILER.begin(d, 0) if PROFILER is not None else None
    if ACCOUNT is not None:
        ACCOUNT.add(d, 0)
    nb = lib().conv2d_workspace_bytes(ctypes.byref(d), 0)
    ws = workspace(nb, "splitk", x.device) if nb else None
    U, variant = xf_cache.get(d, 0, w) if (xf_cache is not None and d.R == 3 and d.S == 3) else (None, -1)
    V, vvar = None, -1
    if keep_input_xf is not None:
        vvar = _shared_input_variant(d)
        if vvar >= 0:
            V = torch.empty(int(lib().conv2d_input_xf_bytes(ctypes.byref(d), vvar)) // 4, dtype=f32, device=x.device)
            keep_input_xf[x.data_ptr()] = (V, vvar)
    lib().conv2d_fwd_keep(ctypes.byref(d), ptr(_chk(x)), ptr(_chk(w)), ptr(bias), ptr(residual),
                          ptr(y), epilogue, ptr(ws), ptr(U), variant, ptr(V), vvar, _stream())
    if t0 is not None:
        PROFILER.end(d, 0, t0)
    return y


def conv2d_dgrad(d, dy, w, residual=None, mask_ref=None, epilogue=0, out=None, xf_cache=None):
    dx = out if out is not None else torch.empty((d.N, d.H, d.W, d.C), dtype=f32, device=dy.device)
    if PROFILER is None:
        def run():                                    # scratch output; no accumulate into it
            tmp = workspace(4 * d.N * d.H * d.W * d.C, "tune_out", dy.device)
            nb_ = lib().conv2d_workspace_bytes(ctypes.byref(d), 1)
            ws_ = workspace(nb_, "splitk", dy.device) if nb_ else None
            lib().conv2d_dgrad(ctypes.byref(d), ptr(_chk_y(d, dy)), ptr(_chk(w)), ptr(residual), ptr(mask_ref), ptr(tmp),
                               epilogue & ~EPI_ACCUM, ptr(ws_), _stream())
        _autotune(d, 1, run)
    t0 = PROFILER.begin(d, 1) if PROFILER is not None else None
    if ACCOUNT is not None:
        ACCOUNT.add(d, 1)
    nb = lib().conv2d_workspace_bytes(ctypes.byref(d), 1)
    ws = workspace(nb, "splitk", dy.device) if nb else None
    U, variant = xf_cache.get(d, 1, w) if (xf_cache is not None and d.R == 3 and d.S == 3) else (None, -1)
    lib().conv2d_dgrad_xf(ctypes.byref(d), ptr(_chk_y(d, dy)), ptr(_chk(w)), ptr(residual), ptr(mask_ref),
                          ptr(dx), epilogue, ptr(ws), ptr(U), variant, _stream())
    if t0 is not None:
        PROFILER.end(d, 1, t0)
    return dx


def conv2d_wgrad(d, x, dy, dw, out_scale=None, dbias=None, beta=0.0, input_xf=None, dbias_scale=None):
    """input_xf: (V, variant) kept by this layer's conv2d_fwd(keep_input_xf=...) for the same x, or None.
    dbias_scale: per-channel factor on the bias gradient (a bias that enters through a folded factor), or None."""
    if PROFILER is None:
        def run():                                    # scratch filter gradient, beta = 0
            tmp = workspace(4 * d.R * d.S * d.C * d.K, "tune_out", x.device)
            nb_ = lib().conv2d_wgrad_workspace_bytes(ctypes.byref(d))
            ws_ = workspace(nb_, "wgrad", x.device)
            lib().conv2d_wgrad(ctypes.byref(d), ptr(_chk(x)), ptr(_chk_y(d, dy)), ptr(out_scale), ptr(tmp), None, 0.0,
                               ptr(ws_), _stream())
        _autotune(d, 2, run)
    nbytes = lib().conv2d_wgrad_workspace_bytes(ctypes.byref(d))
    ws = workspace(nbytes, "wgrad", x.device)
    t0 = PROFILER.begin(d, 2) if PROFILER is not None else None
    if ACCOUNT is not None:
        ACCOUNT.add(d, 2)
    V, vvar = input_xf if input_xf is not None else (None, -1)
    if V is not None:
        V.record_stream(torch.cuda.current_stream())     # made on the forward's stream, read on this one
    lib().conv2d_wgrad_ex(ctypes.byref(d), ptr(_chk(x)), ptr(_chk_y(d, dy)), ptr(out_scale), ptr(dw),
                          ptr(dbias), ptr(dbias_scale), float(beta), ptr(ws), ptr(V), vvar, _stream())
    if t0 is not None:
        PROFILER.end(d, 2, t0)
    return dw


_ptr_tables = {}


def _ptr_table(tensors, device):
    """Device array of the tensors' addresses. Cached by the address tuple: in steady state the caching allocator
    hands every step the same addresses, so no host->device copy (a synchronising one) happens on the step."""
    key = tuple(t.data_ptr() if t is not None else 0 for t in tensors)
    tab = _ptr_tables.get(key)
    if tab is None:
        if len(_ptr_tables) > 4096:
            _ptr_tables.clear()
        tab = torch.tensor(key, dtype=torch.int64, device=device)
        _ptr_tables[key] = tab
    return tab


def conv2d_wgrad_grouped(d, xs, dys, dws, scales=None, beta=0.0):
    """n filter gradients of one descriptor in one launch (mtlssl_conv2d_wgrad_grouped): xs / dys / dws lists of
    tensors, scales a list of per-output-channel scale tensors (or None)."""
    n = len(xs)
    dev = xs[0].device
    for t in list(xs) + list(dys) + list(dws):
        _chk(t)
    if ACCOUNT is not None:
        ACCOUNT.add(d, 2, n)
    nb = lib().conv2d_wgrad_grouped_workspace_bytes(ctypes.byref(d), n)
    ws = workspace(nb, "wgrad_grouped", dev)
    lib().conv2d_wgrad_grouped(ctypes.byref(d), n, ptr(_ptr_table(xs, dev)), ptr(_ptr_table(dys, dev)),
                               ptr(_ptr_table(scales, dev)) if scales is not None else None,
                               ptr(_ptr_table(dws, dev)), float(beta), ptr(ws), _stream())


GROUPED_FWD = os.environ.get("MTLSSL_GROUPED_FWD", "1") != "0"
# problems with more rows than this fill the chip on their own and keep their tuned plans (the grouped launch runs the
# register-staged engine without a K split). Same-box A/B on configs[4] (profiles/r06_grouped_fwd_ab.txt): separate
# launches 108.2 ms/step, grouped up to 5 000 rows 107.9, up to 17 000 rows 107.5, every block 107.5
GROUPED_FWD_MAX_ROWS = int(os.environ.get("MTLSSL_GROUPED_FWD_MAX_ROWS", "20000"))


class _GroupEntry(ctypes.Structure):
    """mtlssl_conv_group_entry."""
    _fields_ = [("w", ctypes.c_void_p), ("y", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("K", ctypes.c_int32),
                ("ldy", ctypes.c_int32), ("epilogue", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def conv2d_fwd_grouped(x, problems):
    """n pointwise forward convolutions on the same input in one launch (mtlssl_conv2d_fwd_grouped).
    problems: list of (desc, w, bias, epilogue, out) — desc of each problem on x (pointwise: 1x1 / stride 1; its ldy set when
    `out` is a channel-slice view), out a tensor / view to write, or None for a fresh dense tensor. -> list of outputs."""
    d0 = problems[0][0]
    outs = []
    for d, w, bias, epi, out in problems:
        assert desc_is_pointwise(d) and (d.N, d.H, d.W, d.C) == (d0.N, d0.H, d0.W, d0.C) and d.C % 16 == 0 \
            and d.K % 4 == 0 and d.K >= 16 and not (epi & EPI_RESIDUAL), (d.C, d.K, epi)
        if out is None:
            assert not d.ldy
            out = torch.empty((d.N, d.OH, d.OW, d.K), dtype=f32, device=x.device)
        else:
            _chk_y(d, out)
        _chk(w)
        outs.append(out)
    ks = [d.K for d, _, _, _, _ in problems]
    # the per-problem records travel as kernel arguments (a HOST array of 3 pointers + 4 int32 each): no device table
    assert len(problems) <= 4, len(problems)
    tab = (_GroupEntry * len(problems))()
    for i, ((d, w, bias, epi, _), out) in enumerate(zip(problems, outs)):
        tab[i] = _GroupEntry(w.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else 0, d.K, int(d.ldy),
                             int(epi), 0)
    dsum = None
    if ACCOUNT is not None or PROFILER is not None:     # booked as ONE problem of the summed width: the same products
        dsum = ConvDesc(d0.N, d0.H, d0.W, d0.C, sum(ks), 1, 1, d0.OH, d0.OW, 1, 1, 0, 0, 0)
    t0 = PROFILER.begin(dsum, 0) if PROFILER is not None else None
    if ACCOUNT is not None:
        ACCOUNT.add(dsum, 0)
    lib().conv2d_fwd_grouped(ctypes.byref(d0), ptr(_chk(x)), len(problems), ctypes.addressof(tab), max(ks), sum(ks), _stream())
    if t0 is not None:
        PROFILER.end(dsum, 0, t0)
    return outs


SEG_DGRAD = os.environ.get("MTLSSL_SEG_DGRAD", "1") != "0"
SEG_DGRAD_MAX_ROWS = int(os.environ.get("MTLSSL_SEG_DGRAD_MAX_ROWS", "1000000000"))


class _SegEntry(ctypes.Structure):
    """mtlssl_conv_seg_entry."""
    _fields_ = [("dy", ctypes.c_void_p), ("w", ctypes.c_void_p), ("K", ctypes.c_int32), ("ldy", ctypes.c_int32)]


def conv2d_dgrad_segmented(segments, residual=None, mask_ref=None, epilogue=0):
    """Input gradient of n pointwise convolutions on the same input in one launch and one accumulator pass
    (mtlssl_conv2d_dgrad_segmented): dx = sum_i dy_i . w_i^T (+ residual, masked).
    segments: list of (desc, dy, w) — desc of each problem on the shared input (pointwise; its ldy set when dy is a
    channel-slice view). -> dx."""
    d0 = segments[0][0]
    assert 1 <= len(segments) <= 4, len(segments)
    tab = (_SegEntry * len(segments))()
    for i, (d, dy, w) in enumerate(segments):
        assert desc_is_pointwise(d) and (d.N, d.H, d.W, d.C) == (d0.N, d0.H, d0.W, d0.C) and d.K % 16 == 0, (d.C, d.K)
        tab[i] = _SegEntry(_chk_y(d, dy).data_ptr(), _chk(w).data_ptr(), d.K, int(d.ldy))
    dx = torch.empty((d0.N, d0.H, d0.W, d0.C), dtype=f32, device=segments[0][1].device)
    dsum = None
    if ACCOUNT is not None or PROFILER is not None:     # booked as ONE problem of the summed width: the same products
        dsum = ConvDesc(d0.N, d0.H, d0.W, d0.C, sum(d.K for d, _, _ in segments), 1, 1, d0.OH, d0.OW, 1, 1, 0, 0, 0)
    t0 = PROFILER.begin(dsum, 1) if PROFILER is not None else None
    if ACCOUNT is not None:
        ACCOUNT.add(dsum, 1)
    lib().conv2d_dgrad_segmented(ctypes.byref(d0), len(segments), ctypes.addressof(tab), ptr(residual), ptr(mask_ref), ptr(dx),
                                 int(epilogue), _stream())
    if t0 is not None:
        PROFILER.end(dsum, 1, t0)
    return dx


def depthwise_fwd(d, x, w, bias=None, epilogue=0):
    y = torch.empty((d.N, d.OH, d.OW, d.K), dtype=f32, device=x.device)
    t0 = _hbm_begin("depthwise_fwd")
    lib().depthwise_fwd(ctypes.byref(d), ptr(_chk(x)), ptr(_chk(w)), ptr(bias), ptr(y), epilogue, _stream())
    if t0 is not None:
        HBM_PROFILER.end("depthwise_fwd", 4 * (x.numel() + y.numel()), t0)
    return y


def depthwise_dgrad(d, dy, w, mask_ref=None, epilogue=0):
    dx = torch.empty((d.N, d.H, d.W, d.C), dtype=f32, device=dy.device)
    t0 = _hbm_begin("depthwise_dgrad")
    lib().depthwise_dgrad(ctypes.byref(d), ptr(_chk(dy)), ptr(_chk(w)), ptr(mask_ref), ptr(dx), epilogue,
                          _stream())
    if t0 is not None:
        HBM_PROFILER.end("depthwise_dgrad", 4 * (dy.numel() + dx.numel() + (dx.numel() if mask_ref is not None else 0)), t0)
    return dx


def depthwise_wgrad(d, x, dy, dw, out_scale=None, beta=0.0):
    ws = workspace(lib().depthwise_wgrad_workspace_bytes(ctypes.byref(d)), "dwgrad", x.device)
    t0 = _hbm_begin("depthwise_wgrad")
    lib().depthwise_wgrad(ctypes.byref(d), ptr(_chk(x)), ptr(_chk(dy)), ptr(out_scale), ptr(dw), float(beta),
                          ptr(ws), _stream())
    if t0 is not None:
        HBM_PROFILER.end("depthwise_wgrad", 4 * (x.numel() + dy.numel()), t0)
    return dw


def bn_param_grads(y, g, gamma, beta_param, dgamma, dbeta, beta=0.0):
    """dgamma/dbeta of a folded inference-mode BatchNorm; `beta` != 0 accumulates."""
    C = y.shape[-1]
    ws = workspace(lib().bn_param_grads_workspace_bytes(int(C)), "bngrad", y.device)
    lib().bn_param_grads(ptr(_chk(y)), ptr(_chk(g)), ptr(gamma), ptr(beta_param), ptr(dgamma), ptr(dbeta),
                         y.numel() // C, int(C), float(beta), ptr(ws), _stream())
    return dgamma, dbeta


def pool_geometry(H, W, k, stride, padding):
    """-> (pad_t, pad_l, OH, OW) of a k x k / stride pooling window under TF 'SAME' / 'VALID'."""
    if padding == "SAME":
        pt, OH = same_pad(H, k, stride)
        pl, OW = same_pad(W, k, stride)
    else:
        pt = pl = 0
        OH, OW = (H - k) // stride + 1, (W - k) // stride + 1
    return pt, pl, OH, OW


def _slice_ld(t, C):
    """Row stride of the channel slice `t` = big[..., c0:c0+C] of a contiguous NHWC map (a torch view)."""
    assert t.is_cuda and t.dtype == f32 and t.dim() == 4 and t.shape[-1] == C and t.stride(-1) == 1 \
        and t.stride(-3) == t.stride(-2) * t.shape[-2], (tuple(t.shape), t.stride())
    return int(t.stride(-2))


def maxpool_fwd(x, k, stride, padding="VALID", out=None):
    """out: a channel-slice view of a wider NHWC map to pool into (the pooling branch of a tf.concat), or None."""
    N, H, W, C = x.shape
    pt, pl, OH, OW = pool_geometry(H, W, k, stride, padding)
    if out is not None:
        assert tuple(out.shape) == (N, OH, OW, C), (tuple(out.shape), (N, OH, OW, C))
        lib().maxpool_fwd_strided(ptr(_chk(x)), ptr(out), N, H, W, C, k, stride, pt, pl, OH, OW, _slice_ld(out, C), _stream())
        return out, (pt, pl)
    y = torch.empty((N, OH, OW, C), dtype=f32, device=x.device)
    lib().maxpool_fwd(ptr(_chk(x)), ptr(y), N, H, W, C, k, stride, pt, pl, OH, OW, _stream())
    return y, (pt, pl)


def maxpool_bwd(x, y, dy, k, stride, pads):
    """y / dy: dense, or channel-slice views (same geometry) of wider NHWC maps."""
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    if not (y.is_contiguous() and dy.is_contiguous()):
        ld = _slice_ld(y, C)
        assert _slice_ld(dy, C) == ld, (ld, dy.stride())
        lib().maxpool_bwd_strided(ptr(x), ptr(y), ptr(dy), ptr(dx), N, H, W, C, k, stride, pads[0], pads[1],
                                  y.shape[1], y.shape[2], ld, _stream())
        return dx
    lib().maxpool_bwd(ptr(x), ptr(y), ptr(_chk(dy)), ptr(dx), N, H, W, C, k, stride, pads[0],
                      pads[1], y.shape[1], y.shape[2], _stream())
    return dx


def avgpool_fwd(x, k, stride, padding="SAME"):
    N, H, W, C = x.shape
    if padding == "SAME":
        pt, OH = same_pad(H, k, stride)
        pl, OW = same_pad(W, k, stride)
    else:
        pt = pl = 0
        OH, OW = (H - k) // stride + 1, (W - k) // stride + 1
    y = torch.empty((N, OH, OW, C), dtype=f32, device=x.device)
    lib().avgpool_fwd(ptr(_chk(x)), ptr(y), N, H, W, C, k, stride, pt, pl, OH, OW, _stream())
    return y, (pt, pl)


def avgpool_bwd(dy, x_shape, k, stride, pads):
    N, H, W, C = x_shape
    dx = torch.empty(tuple(x_shape), dtype=f32, device=dy.device)
    lib().avgpool_bwd(ptr(_chk(dy)), ptr(dx), N, H, W, C, k, stride, pads[0], pads[1], dy.shape[1], dy.shape[2],
                      _stream())
    return dx


def concat_channels(parts):
    """tf.concat(axis=3) of NHWC tensors."""
    C = sum(int(t.shape[-1]) for t in parts)
    out = torch.empty(tuple(parts[0].shape[:-1]) + (C,), dtype=f32, device=parts[0].device)
    rows = out.numel() // C
    c0 = 0
    for t in parts:
        nc = int(t.shape[-1])
        lib().copy_channels(ptr(_chk(t)), nc, 0, ptr(out), C, c0, rows, nc, 0, _stream())
        c0 += nc
    return out


def slice_channels(t, c0, nc):
    """t[..., c0:c0+nc] as a contiguous tensor."""
    C = int(t.shape[-1])
    out = torch.empty(tuple(t.shape[:-1]) + (nc,), dtype=f32, device=t.device)
    lib().copy_channels(ptr(_chk(t)), C, c0, ptr(out), nc, 0, t.numel() // C, nc, 0, _stream())
    return out


def spatial_mean_fwd(x):
    N, H, W, C = x.shape
    y = torch.empty((N, C), dtype=f32, device=x.device)
    lib().spatial_mean_fwd(ptr(_chk(x)), ptr(y), N, H * W, C, _stream())
    return y


def spatial_mean_bwd(dy, shape, mask_ref=None, mask6=False):
    """mask_ref: the averaged activation; when given the ReLU (ReLU6) gradient is fused in."""
    N, H, W, C = shape
    dx = torch.empty(tuple(shape), dtype=f32, device=dy.device)
    if mask_ref is not None and C % 4 == 0:
        lib().spatial_mean_bwd_masked(ptr(_chk(dy)), ptr(_chk(mask_ref)), ptr(dx), N, H * W, C, 1 if mask6 else 0,
                                      _stream())
        return dx
    lib().spatial_mean_bwd(ptr(_chk(dy)), ptr(dx), N, H * W, C, _stream())
    if mask_ref is not None:
        return (relu6_bwd if mask6 else relu_bwd)(mask_ref, dx, out=dx)
    return dx


# ------------------------------------------------------------------------------- detection family
def anchors_generate(grid_h, grid_w, scales, aspect_ratios, base=(256.0, 256.0),
                     stride=(16.0, 16.0), offset=(0.0, 0.0), device="cuda"):
    sc = (ctypes.c_float * len(scales))(*scales)
    ar = (ctypes.c_float * len(aspect_ratios))(*aspect_ratios)
    n = grid_h * grid_w * len(scales) * len(aspect_ratios)
    out = torch.empty((n, 4), dtype=f32, device=device)
    lib().anchors_generate(ptr(out), grid_h, grid_w, ctypes.cast(sc, ctypes.c_void_p), len(scales),
                           ctypes.cast(ar, ctypes.c_void_p), len(aspect_ratios), base[0], base[1],
                           stride[0], stride[1], offset[0], offset[1], _stream())
    return out


def prune_outside_window(boxes, window):
    n = boxes.shape[0]
    keep = torch.empty((n,), dtype=i32, device=boxes.device)
    cnt = torch.zeros((1,), dtype=i32, device=boxes.device)
    lib().boxes_prune_outside_window(ptr(_chk(boxes)), n, float(window[0]), float(window[1]),
                                     float(window[2]), float(window[3]), ptr(keep), ptr(cnt),
                                     _stream())
    return keep[: int(cnt.item())]       # one-time host sync (anchors are static per input size)


def clip_to_window(boxes, window):
    out = torch.empty_like(boxes)
    lib().boxes_clip_to_window(ptr(_chk(boxes)), boxes.shape[0], float(window[0]), float(window[1]),
                               float(window[2]), float(window[3]), ptr(out), _stream())
    return out


def gather_rows(src, idx):
    B, n_src, L = src.shape
    out = torch.empty((B, idx.numel(), L), dtype=f32, device=src.device)
    lib().gather_rows(ptr(_chk(src)), ptr(_chk(idx, i32)), ptr(out), B, n_src, idx.numel(), L,
                      _stream())
    return out


def scatter_rows(src, idx, n_dst):
    B, n_idx, L = src.shape
    out = torch.zeros((B, n_dst, L), dtype=f32, device=src.device)
    lib().scatter_rows(ptr(_chk(src)), ptr(_chk(idx, i32)), ptr(out), B, n_dst, n_idx, L, _stream())
    return out


def boxes_decode(codes, anchors, scale_factors=(10.0, 10.0, 5.0, 5.0)):
    B, n, _ = codes.shape
    out = torch.empty_like(codes)
    lib().boxes_decode(ptr(_chk(codes)), ptr(_chk(anchors)), ptr(out), B, n,
                       1 if anchors.dim() == 3 else 0, *[float(s) for s in scale_factors], _stream())
    return out


def boxes_encode(boxes, anchors, scale_factors=(10.0, 10.0, 5.0, 5.0)):
    out = torch.empty_like(boxes)
    lib().boxes_encode(ptr(_chk(boxes)), ptr(_chk(anchors)), ptr(out), boxes.shape[0],
                       *[float(s) for s in scale_factors], _stream())
    return out


def rpn_proposals(enc, logits, anchors, img_h, img_w, score_thresh, iou_thresh, max_proposals):
    B, n, _ = enc.shape
    dev = enc.device
    ws = workspace(lib().rpn_proposals_workspace_bytes(B, n), "nms", dev)
    boxes = torch.empty((B, max_proposals, 4), dtype=f32, device=dev)
    scores = torch.empty((B, max_proposals), dtype=f32, device=dev)
    num = torch.empty((B,), dtype=i32, device=dev)
    lib().rpn_proposals(ptr(_chk(enc)), ptr(_chk(logits)), ptr(_chk(anchors)), B, n, float(img_h),
                        float(img_w), float(score_thresh), float(iou_thresh), max_proposals,
                        ptr(boxes), ptr(scores), ptr(num), ptr(ws), _stream())
    return boxes, scores, num


def nms(boxes, scores, iou_thresh, max_out):
    n = boxes.shape[0]
    dev = boxes.device
    ws = workspace(lib().nms_workspace_bytes(max(n, 1)), "nms", dev)
    sel = torch.empty((max_out,), dtype=i32, device=dev)
    num = torch.empty((1,), dtype=i32, device=dev)
    lib().nms(ptr(_chk(boxes)), ptr(_chk(scores)), n, float(iou_thresh), max_out, ptr(sel), ptr(num),
              ptr(ws), _stream())
    return sel, num


def batch_multiclass_nms(boxes, scores, score_thresh, iou_thresh, max_per_class, max_total,
                         clip_window=None, change_coordinate_frame=False, num_valid=None, col0=0,
                         num_classes=None):
    """core/post_processing.py:167-312. boxes [B,n,q,4], scores [B,n,Cfull]; classes are columns
    col0 .. col0+num_classes-1 of `scores` -> (boxes [B,T,4], scores [B,T], classes [B,T] float,
    num int32[B])."""
    B, n, q, _ = boxes.shape
    ld = scores.shape[2]
    C = int(num_classes) if num_classes is not None else ld - col0
    dev = boxes.device
    T = int(max_total)
    ob = torch.empty((B, T, 4), dtype=f32, device=dev)
    os_ = torch.empty((B, T), dtype=f32, device=dev)
    oc = torch.empty((B, T), dtype=f32, device=dev)
    on = torch.empty((B,), dtype=i32, device=dev)
    ws = workspace(lib().batch_multiclass_nms_workspace_bytes(B, max(n, 1), C, int(max_per_class)), "mcnms", dev)
    win = (ctypes.c_float * 4)(*[float(v) for v in clip_window]) if clip_window is not None else None
    lib().batch_multiclass_nms(ptr(_chk(boxes)), ptr(_chk(scores)) + 4 * col0, ld, ptr(num_valid), B, n, q, C,
                               float(score_thresh), float(iou_thresh), int(max_per_class), T, win,
                               1 if change_coordinate_frame else 0, ptr(ob), ptr(os_), ptr(oc), ptr(on),
                               ptr(ws), _stream())
    return ob, os_, oc, on


def hard_example_mining(loc_rl, cls_rl, boxes, num_proposals, d_box, d_cls, num_hard_examples, iou_threshold, loss_type):
    """core/losses.py:418-631 on the second stage (see mtlssl_hard_mining_*): loc_rl / cls_rl [B,n2] per-proposal
    losses, boxes [B,n2,4] the proposal boxes. Zeroes the gradient rows of proposals that were not mined (in place)
    and returns (mined loc loss [B], mined cls loss [B], selected [B,max_sel], number kept [B]). The NMS sees padding
    rows (score -inf) LAST, so they never suppress a proposal; the ones it may append are dropped by _apply, which also
    returns the count without them (= NMS over the first num_proposals rows, the reference's unpad step)."""
    B, n2 = loc_rl.shape
    dev = loc_rl.device
    scores = torch.empty((B, n2), dtype=f32, device=dev)
    lib().hard_mining_scores(ptr(_chk(loc_rl)), ptr(_chk(cls_rl)), ptr(_chk(num_proposals, i32)), B, n2,
                             {"both": 0, "cls": 1, "loc": 2}[loss_type], ptr(scores), _stream())
    max_sel = int(num_hard_examples) if num_hard_examples else n2
    sel = torch.empty((B, max_sel), dtype=i32, device=dev)
    num = torch.empty((B,), dtype=i32, device=dev)
    ws = workspace(lib().nms_workspace_bytes(max(n2, 1)), "nms", dev)
    for b in range(B):                       # per image, like the reference's loop over the clone's images
        lib().nms(ptr(_chk(boxes[b])), ptr(scores[b]), n2, float(iou_threshold), max_sel, ptr(sel[b]), ptr(num[b:b + 1]),
                  ptr(ws), _stream())
    loc_loss = torch.empty((B,), dtype=f32, device=dev)
    cls_loss = torch.empty((B,), dtype=f32, device=dev)
    kept = torch.empty((B,), dtype=i32, device=dev)
    lib().hard_mining_apply(ptr(sel), ptr(num), ptr(_chk(num_proposals, i32)), B, max_sel, n2, ptr(loc_rl), ptr(cls_rl),
                            ptr(_chk(d_box)), d_box.numel() // (B * n2), ptr(_chk(d_cls)), d_cls.numel() // (B * n2),
                            ptr(loc_loss), ptr(cls_loss), ptr(kept), _stream())
    return loc_loss, cls_loss, sel, kept


def dropout(x, keep_prob, seed, stream_id, out=None):
    """slim.dropout in training mode with counter-hash draws (mtlssl_dropout). Call it again on the gradient with the
    same (seed, stream_id) for the backward pass."""
    y = out if out is not None else torch.empty_like(x)
    lib().dropout(ptr(_chk(x)), ptr(y), x.numel(), float(keep_prob), int(seed) & 0xFFFFFFFF, int(stream_id) & 0xFFFFFFFF,
                  _stream())
    return y


def score_convert(logits, mode):
    """mode: 'SOFTMAX' | 'SIGMOID' | 'IDENTITY' (builders/post_processing_builder.py:80-108)."""
    if mode == "IDENTITY":
        return logits
    out = torch.empty_like(logits)
    C = logits.shape[-1]
    lib().score_convert(ptr(_chk(logits)), ptr(out), logits.numel() // C, C, {"SOFTMAX": 1, "SIGMOID": 2}[mode],
                        _stream())
    return out


def assign_targets(anchors, gt_boxes, num_gt, gt_labels, unmatched_cls_target, matched_thresh,
                   unmatched_thresh, force_match, gt_extra=None, want=("match", "cls_targets",
                   "cls_weights", "reg_targets", "reg_weights")):
    """anchors [n,4] or [B,n,4]; gt_boxes [B,G,4]; num_gt int32[B]; gt_labels [B,G,d] or None."""
    dev = gt_boxes.device
    B, G, _ = gt_boxes.shape
    batched = anchors.dim() == 3
    n = anchors.shape[-2]
    d = 1 if gt_labels is None else gt_labels.shape[-1]
    e = 0 if gt_extra is None else gt_extra.shape[-1]
    out = {"match": torch.empty((B, n), dtype=i32, device=dev)}
    if "cls_targets" in want:
        out["cls_targets"] = torch.empty((B, n, d), dtype=f32, device=dev)
    if "cls_weights" in want:
        out["cls_weights"] = torch.empty((B, n), dtype=f32, device=dev)
    if "reg_targets" in want:
        out["reg_targets"] = torch.empty((B, n, 4), dtype=f32, device=dev)
    if "reg_weights" in want:
        out["reg_weights"] = torch.empty((B, n), dtype=f32, device=dev)
    if gt_extra is not None:
        out["extra_targets"] = torch.empty((B, n, e), dtype=f32, device=dev)
    ws = workspace(lib().assign_targets_workspace_bytes(B, n, G), "assign", dev)
    lib().assign_targets(ptr(_chk(anchors)), int(batched), B, n, ptr(_chk(gt_boxes)),
                         ptr(_chk(num_gt, i32)), G, ptr(gt_labels), d, ptr(gt_extra), e,
                         ptr(unmatched_cls_target), float(matched_thresh), float(unmatched_thresh),
                         int(force_match), ptr(out["match"]), ptr(out.get("cls_targets")),
                         ptr(out.get("cls_weights")), ptr(out.get("reg_targets")),
                         ptr(out.get("reg_weights")), ptr(out.get("extra_targets")), ptr(ws),
                         _stream())
    return out


def balanced_sample(indicator, labels, batch_size, positive_fraction, seed, stream_id0,
                    stream_stride):
    B, n = indicator.shape
    out = torch.empty_like(indicator)
    lib().balanced_sample(ptr(_chk(indicator)), ptr(_chk(labels)), B, n, batch_size,
                          float(positive_fraction), seed, stream_id0, stream_stride, ptr(out),
                          _stream())
    return out


def sample_proposals(proposals, num_proposals, gt_boxes, num_gt, gt_labels_bg, n2,
                     balance_fraction, seed, stream_id0, stream_stride, img_h, img_w):
    B, max_p, _ = proposals.shape
    dev = proposals.device
    G, d = gt_labels_bg.shape[1], gt_labels_bg.shape[2]
    ob = torch.empty((B, n2, 4), dtype=f32, device=dev)
    on = torch.empty((B, n2, 4), dtype=f32, device=dev)
    num = torch.empty((B,), dtype=i32, device=dev)
    lib().sample_proposals(ptr(_chk(proposals)), ptr(_chk(num_proposals, i32)), B, max_p,
                           ptr(_chk(gt_boxes)), ptr(_chk(num_gt, i32)), G, ptr(_chk(gt_labels_bg)),
                           d, n2, float(balance_fraction), seed, stream_id0, stream_stride,
                           float(img_h), float(img_w), ptr(ob), ptr(on), ptr(num), _stream())
    return ob, on, num


def roi_crop_pool_fwd(feat, boxes, box_ind, crop, pool_k=1, pool_stride=1, want_argmax=True):
    B, H, W, C = feat.shape
    R = boxes.shape[0]
    P = (crop - pool_k) // pool_stride + 1
    out = torch.empty((R, P, P, C), dtype=f32, device=feat.device)
    am = (torch.empty((R, P, P, C), dtype=torch.uint8, device=feat.device)
          if (want_argmax and pool_k > 1) else None)
    t0 = _hbm_begin("roi_crop_pool_fwd")
    lib().roi_crop_pool_fwd(ptr(_chk(feat)), B, H, W, C, ptr(_chk(boxes)), ptr(_chk(box_ind, i32)),
                            R, crop, pool_k, pool_stride, ptr(out), ptr(am), _stream())
    if t0 is not None:
        HBM_PROFILER.end("roi_crop_pool_fwd", 4 * (feat.numel() + out.numel()) + (am.numel() if am is not None else 0), t0)
    return out, am


def roi_crop_pool_bwd(dout, argmax, feat_shape, boxes, box_ind, crop, pool_k, pool_stride,
                      dfeat=None, accumulate=None, algo=0):
    """dfeat (+)= gradient of roi_crop_pool_fwd. Without `dfeat` a fresh map is returned (written in full by the
    kernel, no memset); with one, the gradient is added to it unless accumulate=False (then it is overwritten).
    algo: 0 auto, 1 the LDS-resident deterministic kernel, 2 HBM atomics (mtlssl_roi_crop_pool_bwd_ex)."""
    B, H, W, C = feat_shape
    if dfeat is None:
        dfeat = torch.empty(feat_shape, dtype=f32, device=dout.device)
        accumulate = False
    elif accumulate is None:
        accumulate = True
    ws = workspace(lib().roi_crop_pool_bwd_workspace_bytes(), "roi_bwd", dout.device)
    lib().roi_crop_pool_bwd_ex(ptr(_chk(dout)), ptr(argmax), B, H, W, C, ptr(boxes), ptr(box_ind),
                               boxes.shape[0], crop, pool_k, pool_stride, ptr(dfeat), 1 if accumulate else 0,
                               int(algo), ptr(ws), _stream())
    return dfeat


def psroi_fwd(fmap, boxes, box_ind, crop, bins):
    B, H, W, C = fmap.shape
    R = boxes.shape[0]
    Cc = C // max(bins[0] * bins[1], 1)     # invalid geometry is rejected by the C ABI below
    out = torch.empty((R, Cc), dtype=f32, device=fmap.device)
    t0 = _hbm_begin("psroi_fwd")
    lib().psroi_fwd(ptr(_chk(fmap)), B, H, W, C, ptr(_chk(boxes)), ptr(_chk(box_ind, i32)), R, crop[0],
                    crop[1], bins[0], bins[1], ptr(out), _stream())
    if t0 is not None:
        HBM_PROFILER.end("psroi_fwd", 4 * (fmap.numel() + out.numel()), t0)
    return out


def psroi_bwd(dout, fmap_shape, boxes, box_ind, crop, bins, dfmap=None):
    B, H, W, C = fmap_shape
    if dfmap is None:
        dfmap = torch.zeros(fmap_shape, dtype=f32, device=dout.device)
    t0 = _hbm_begin("psroi_bwd")
    lib().psroi_bwd(ptr(_chk(dout)), B, H, W, C, ptr(_chk(boxes)), ptr(_chk(box_ind, i32)), boxes.shape[0],
                    crop[0], crop[1], bins[0], bins[1], ptr(dfmap), _stream())
    if t0 is not None:
        HBM_PROFILER.end("psroi_bwd", 4 * (dout.numel() + dfmap.numel()), t0)
    return dfmap


def resize_bilinear_fwd(x, OH, OW):
    N, H, W, C = x.shape
    y = torch.empty((N, OH, OW, C), dtype=f32, device=x.device)
    lib().resize_bilinear_fwd(ptr(_chk(x)), ptr(y), N, H, W, C, OH, OW, _stream())
    return y


def resize_bilinear_bwd(dy, in_shape):
    N, H, W, C = in_shape
    dx = torch.zeros(in_shape, dtype=f32, device=dy.device)
    lib().resize_bilinear_bwd(ptr(_chk(dy)), ptr(dx), N, H, W, C, dy.shape[1], dy.shape[2], _stream())
    return dx


# ------------------------------------------------------------------------------- losses / optimizer
def smooth_l1(pred, target, row_scale, sigma, want_grad=True):
    rows, cs = pred.numel() // pred.shape[-1], pred.shape[-1]
    rl = torch.empty((rows,), dtype=f32, device=pred.device)
    dp = torch.empty_like(pred) if want_grad else None
    lib().smooth_l1_fwd_bwd(ptr(_chk(pred)), ptr(_chk(target)), ptr(row_scale), rows, cs,
                            float(sigma), ptr(rl), ptr(dp), _stream())
    return rl, dp


def softmax_ce(logits, targets, row_scale=None, col0=0, C=None, want_grad=True, dlogits=None):
    ldl, ldt = logits.shape[-1], targets.shape[-1]
    rows = logits.numel() // ldl
    C = C if C is not None else ldl - col0
    rl = torch.empty((rows,), dtype=f32, device=logits.device)
    if want_grad and dlogits is None:
        dlogits = torch.zeros_like(logits) if (col0 or C != ldl) else torch.empty_like(logits)
    lib().softmax_ce_fwd_bwd(ptr(_chk(logits)), ldl, ptr(_chk(targets)), ldt, col0, C, ptr(row_scale),
                             rows, ptr(rl), ptr(dlogits), _stream())
    return rl, dlogits


def reduce_sum(x, scale=1.0, out=None):
    out = out if out is not None else torch.empty((1,), dtype=f32, device=x.device)
    lib().reduce_sum(ptr(_chk(x)), x.numel(), float(scale), ptr(out), _stream())
    return out


def sgd_momentum_clip(weights, grads, accum, var_offsets, max_var_size, lr, momentum, clip_norm,
                      grad_scale=1.0, var_weight_decay=None, var_grad_mult=None, fold=None, zero_grads=False):
    """fold: a ParamStore whose shadow weights (ops.fold_scales) the same launch refreshes — only when no scale vector
    depends on a variable being updated (every BatchNorm frozen) — or the triple (eff, fold_ptrs, fold_len) of such a
    store, sliced like weights / var_offsets (a per-bucket update: var_offsets relative to the slice's first element).
    zero_grads: the launch leaves zeros in `grads`."""
    nv = var_offsets.numel() - 1
    norms = workspace(lib().sgd_workspace_bytes(max(nv, 1), int(max_var_size)), "norms", weights.device)
    if fold is not None and not isinstance(fold, tuple):
        fold = (fold.eff, fold.fold_ptrs, fold.fold_len) if fold.eff is not None else None
    if fold is not None or zero_grads:
        eff, fptr, flen = fold if fold is not None else (None, None, None)
        lib().sgd_momentum_clip_fold(ptr(_chk(weights)), ptr(_chk(grads)), ptr(_chk(accum)),
                                     ptr(_chk(var_offsets, i32)), nv, weights.numel(), int(max_var_size),
                                     float(lr), float(momentum), float(clip_norm), float(grad_scale),
                                     ptr(var_weight_decay), ptr(var_grad_mult), ptr(norms),
                                     ptr(eff), ptr(fptr), ptr(flen), 1 if zero_grads else 0, _stream())
        return
    lib().sgd_momentum_clip(ptr(_chk(weights)), ptr(_chk(grads)), ptr(_chk(accum)),
                            ptr(_chk(var_offsets, i32)), nv, weights.numel(), int(max_var_size),
                            float(lr), float(momentum), float(clip_norm), float(grad_scale),
                            ptr(var_weight_decay), ptr(var_grad_mult), ptr(norms), _stream())


def adaptive_update_clip(kind, weights, grads, slot0, slot1, var_offsets, max_var_size, lr, p0, p1, p2, clip_norm,
                         grad_scale=1.0, var_weight_decay=None, var_grad_mult=None):
    """kind 1: RMSProp (slot0 mean square, slot1 momentum; p0 decay, p1 momentum, p2 epsilon);
    kind 2: Adam (slot0 m, slot1 v; p0 beta1, p1 beta2, p2 epsilon; lr already bias-corrected)."""
    nv = var_offsets.numel() - 1
    norms = workspace(lib().sgd_workspace_bytes(max(nv, 1), int(max_var_size)), "norms", weights.device)
    lib().adaptive_update_clip(int(kind), ptr(_chk(weights)), ptr(_chk(grads)), ptr(_chk(slot0)), ptr(_chk(slot1)),
                               ptr(_chk(var_offsets, i32)), nv, weights.numel(), int(max_var_size), float(lr),
                               float(p0), float(p1), float(p2), float(clip_norm), float(grad_scale),
                               ptr(var_weight_decay), ptr(var_grad_mult), ptr(norms), _stream())


def fold_scales(ps):
    """Refresh ParamStore.eff (all registered BatchNorm / residual-scale folds) in one launch."""
    if ps.eff is None:
        return
    lib().fold_scales(ptr(ps.weights), ptr(ps.eff), ptr(ps.var_offsets), len(ps.trainable_specs),
                      ps.weights.numel(), ptr(ps.fold_ptrs), ptr(ps.fold_len), _stream())


class BnRefreshTable:
    """Device pointer tables for `bn_refresh`: built once from the layers whose BatchNorm parameters train
    (the tensors they point to live as long as the model)."""

    def __init__(self, layers, device):
        self.layers = [l for l in layers if getattr(l, "bn_trainable", False)]
        self.n = len(self.layers)
        if not self.n:
            return
        ps = self.layers[0].ps

        def table(fn):
            return torch.tensor([fn(l) for l in self.layers], dtype=torch.int64, device=device)
        self.gamma = table(lambda l: ps.value(l.gamma.name).data_ptr() if l.gamma is not None else 0)
        self.beta = table(lambda l: ps.value(l.beta.name).data_ptr())
        self.mean = table(lambda l: ps.value(l.mean.name).data_ptr())
        self.inv_std = table(lambda l: l.inv_std.data_ptr())
        self.scale = table(lambda l: _chk(l.scale).data_ptr())
        self.shift = table(lambda l: _chk(l.shift).data_ptr())
        ch = [int(l.scale.numel()) for l in self.layers]
        self.channels = torch.tensor(ch, dtype=i32, device=device)
        self.max_c = max(ch)
        self._keep = [(l.inv_std, l.scale, l.shift) for l in self.layers]


def bn_refresh(tab):
    """scale = gamma * inv_std, shift = beta - mean * scale for every trainable-BatchNorm layer, one launch."""
    if tab.n:
        lib().bn_refresh(tab.n, ptr(tab.gamma), ptr(tab.beta), ptr(tab.mean), ptr(tab.inv_std), ptr(tab.scale),
                         ptr(tab.shift), ptr(tab.channels), tab.max_c, _stream())


def axpby(x, y, a, b):
    lib().axpby(ptr(_chk(x)), ptr(_chk(y)), x.numel(), float(a), float(b), _stream())
    return y


def scale_channels(w, scale, out=None):
    out = out if out is not None else torch.empty_like(w)
    K = w.shape[-1]
    lib().scale_channels(ptr(_chk(w)), ptr(_chk(scale)), ptr(out), w.numel() // K, K, _stream())
    return out


def tanh_bwd(y, dy):
    dx = torch.empty_like(dy)
    lib().tanh_bwd(ptr(_chk(y)), ptr(_chk(dy)), ptr(dx), y.numel(), _stream())
    return dx


def relu_bwd(y, dy, out=None):
    dx = out if out is not None else torch.empty_like(dy)
    lib().relu_bwd(ptr(_chk(y)), ptr(_chk(dy)), ptr(dx), y.numel(), _stream())
    return dx


def relu6_bwd(y, dy, out=None):
    dx = out if out is not None else torch.empty_like(dy)
    lib().relu6_bwd(ptr(_chk(y)), ptr(_chk(dy)), ptr(dx), y.numel(), _stream())
    return dx


def bias_add_channels(x, bias):
    out = torch.empty_like(x)
    C = x.shape[-1]
    lib().bias_add_channels(ptr(_chk(x)), ptr(_chk(bias)), ptr(out), x.numel() // C, C, _stream())
    return out


def onehot2(t):
    out = torch.empty(tuple(t.shape) + (2,), dtype=f32, device=t.device)
    lib().onehot2(ptr(_chk(t)), ptr(out), t.numel(), _stream())
    return out


# ------------------------------------------------------------------------------- loss glue
def rpn_loss_scales(sampled, reg_w, loc_coef, obj_coef):
    B, n = sampled.shape
    ls, os_ = torch.empty_like(sampled), torch.empty_like(sampled)
    lib().rpn_loss_scales(ptr(_chk(sampled)), ptr(_chk(reg_w)), B, n, float(loc_coef), float(obj_coef),
                          ptr(ls), ptr(os_), _stream())
    return ls, os_


def detector_loss_scales(cls_w, reg_w, num_proposals, closeness_targets, cls_coef, loc_coef, clo_coef):
    B, n2 = cls_w.shape
    k1 = closeness_targets.shape[-1] if closeness_targets is not None else 0
    cs, ls = torch.empty_like(cls_w), torch.empty_like(cls_w)
    qs = torch.empty_like(cls_w) if closeness_targets is not None else None
    lib().detector_loss_scales(ptr(_chk(cls_w)), ptr(_chk(reg_w)), ptr(_chk(num_proposals, i32)),
                               ptr(closeness_targets), B, n2, k1, float(cls_coef), float(loc_coef),
                               float(clo_coef), ptr(cs), ptr(ls), ptr(qs), _stream())
    return cs, ls, qs


def box_select_smooth_l1(refined, cls_targets, reg_targets, row_scale, sigma=1.0, want_grad=True):
    rows, K = refined.shape[0], refined.shape[1]
    rl = torch.empty((rows,), dtype=f32, device=refined.device)
    dr = torch.empty_like(refined) if want_grad else None
    lib().box_select_smooth_l1(ptr(_chk(refined)), ptr(_chk(cls_targets)), ptr(_chk(reg_targets)),
                               ptr(_chk(row_scale)), rows, K, float(sigma), ptr(rl), ptr(dr), _stream())
    return rl, dr


def edgemask_targets(gt, coef):
    B, _, H, W = gt.shape
    tgt = torch.empty((B, H, W, 2), dtype=f32, device=gt.device)
    sc = torch.empty((B, H, W), dtype=f32, device=gt.device)
    lib().edgemask_targets(ptr(_chk(gt)), B, H, W, float(coef), ptr(tgt), ptr(sc), _stream())
    return tgt, sc


def expand_windows(proposals_norm, n_expand=5):
    B, n2, _ = proposals_norm.shape
    out = torch.empty((B, n_expand, n2, 4), dtype=f32, device=proposals_norm.device)
    lib().expand_windows(ptr(_chk(proposals_norm)), B, n2, n_expand, ptr(out), _stream())
    return out


def dedup_windows(windows, capacity, overflow):
    """windows [B,E,n2,4] -> (rois [B,(E-1)*n2+capacity,4], src_row int32 [B*E*n2]); see mtlssl_dedup_windows."""
    B, E, n2, _ = windows.shape
    rois = torch.empty((B, (E - 1) * n2 + capacity, 4), dtype=f32, device=windows.device)
    src = torch.empty((B * E * n2,), dtype=i32, device=windows.device)
    lib().dedup_windows(ptr(_chk(windows)), B, E, n2, int(capacity), ptr(rois), ptr(src), ptr(_chk(overflow, i32)),
                        _stream())
    return rois, src


def refine_concat(cls, win, clo, B, n2, n_expand, global_closeness):
    k1 = cls.shape[-1]
    ld = k1 + (n_expand * k1 if win is not None else 0) + (k1 if clo is not None else 0)
    out = torch.empty((B * n2, ld), dtype=f32, device=cls.device)
    lib().refine_concat(ptr(_chk(cls)), ptr(win), ptr(clo), B, n2, k1, n_expand, int(global_closeness),
                        ptr(out), _stream())
    return out
