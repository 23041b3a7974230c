"""Per-layer-shape timing of a configuration's convolution calls (HIP events around every conv call, every side stream
off so that each call has the chip to itself): which (mode, shape) classes the step spends its time in, at what rate.
    python tools/layer_times.py <config name> [steps]      config: resnet101 | rfcn | mobilenet | inception
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MTLSSL_AUX_STREAM", "0")
os.environ.setdefault("MTLSSL_WGRAD_STREAM", "0")
os.environ.setdefault("MTLSSL_SPLIT_LOSS", "0")
import torch  # noqa: E402

CFG = {"resnet101": ("frcnn_resnet101_coco_mtl.config", 600, 1024), "rfcn": ("rfcn_resnet101_voc_mtl.config", 600, 1024),
       "mobilenet": ("frcnn_mobilenet_v1_voc_mtl.config", 600, 1024),
       "inception": ("frcnn_inception_resnet_v2_coco_mtl.config", 800, 1333)}


def main():
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, model_builder, ops, synthetic, trainer
    name = sys.argv[1] if len(sys.argv) > 1 else "inception"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    fn, H, W = CFG[name]
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", fn)).read())
    B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
    model = model_builder.build(cfg.model, True, "cuda", seed=0)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    ring = [tr.stage_batch(synthetic.make_batch(B, H, W, K, seed=1234 + 1000 * i, device="cuda")) for i in range(4)]
    for i in range(6):
        tr.step(ring[i % 4])
    torch.cuda.synchronize()

    class P(ops.ConvProfiler):
        def __init__(self):
            super().__init__(None)
            self.rows = []

        def end(self, d, mode, start):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            key = (mode, d.N * d.OH * d.OW if mode == 0 else d.N * d.H * d.W, d.C, d.K, d.R, d.S, d.stride, d.H, d.W)
            self.rows.append((key, start, e, lib_cfg(d, mode), int(ops.lib().conv2d_num_dispatches(ctypes.byref(d), mode)),
                              2.0 * d.N * d.OH * d.OW * d.K * d.C * d.R * d.S))

    def lib_cfg(d, mode):
        return int(ops.lib().conv2d_tile_config(ctypes.byref(d), mode))

    ops.PROFILER = p = P()
    import time
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(ring[i % 4])
    torch.cuda.synchronize()
    ms_step = 1e3 * (time.perf_counter() - t0) / steps
    ops.PROFILER = None
    agg = {}
    for key, s, e, cfgc, nd, fl in p.rows:
        r = agg.setdefault(key, [0, 0.0, 0.0, cfgc, nd])
        r[0] += 1
        r[1] += s.elapsed_time(e)
        r[2] += fl
    tot = sum(r[1] for r in agg.values())
    print("%s: %.2f ms/step serialised with per-call events; conv calls %.2f ms/step in %d calls/step, %d shapes" % (
        name, ms_step, tot / steps, len(p.rows) // steps, len(agg)))
    print("| mode | rows (M) | C | K | RxS/stride | map | plan | launches | calls/step | us/call | ms/step | TFLOP/s (direct) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for key, r in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "45"))]:
        mode, M, C, Kk, R, S, st, Hh, Ww = key
        print("| %s | %d | %d | %d | %dx%d/%d | %dx%d | %d | %d | %.1f | %.1f | %.3f | %.1f |" % (
            ("fwd", "dgrad", "wgrad")[mode], M, C, Kk, R, S, st, Hh, Ww, r[3], r[4], r[0] / steps, 1e3 * r[1] / r[0], r[1] / steps,
            r[2] / (r[1] * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
