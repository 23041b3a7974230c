"""Un-profiled phase breakdown of the training step: HIP events on the main stream at the phase boundaries
(ops.mark) plus the host clock at the same points. Prints, per phase, the main stream's time and how far the
launch thread runs ahead of the GPU — the check that the step is not launch-bound (rocprofv3 slows the launch
thread, so gaps seen in a kernel trace can be the profiler's own).
Usage: python tools/phase_times.py [--steps 12] [--config ...]"""
import argparse
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config"))
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1024)
    a = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    ge.build()
    from mtl_ssl_amd import config, model_builder, ops, synthetic, trainer
    dev = torch.device("cuda", 0)
    cfg = config.parse_pipeline_config(open(a.config).read())
    B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
    model = model_builder.build(cfg.model, True, dev, seed=0)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = tr.stage_batch(synthetic.make_batch(B, a.height, a.width, K, seed=1234, device=dev))
    for _ in range(a.warmup):
        tr.step(batch)
    torch.cuda.synchronize()
    marks = []

    def hook(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, time.perf_counter(), e))
    ops.PHASE_HOOK = hook
    t0 = time.perf_counter()
    hook("origin")
    for _ in range(a.steps):
        tr.step(batch)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    ops.PHASE_HOOK = None
    e0, h0 = marks[0][2], marks[0][1]
    gpu = [e0.elapsed_time(m[2]) for m in marks]              # ms since origin, on the main stream
    host = [1e3 * (m[1] - h0) for m in marks]
    per, lead = collections.defaultdict(list), collections.defaultdict(list)
    for i in range(1, len(marks)):
        per[marks[i][0]].append(gpu[i] - gpu[i - 1])
        lead[marks[i][0]].append(gpu[i] - host[i])
    print("%d steps: %.2f ms/step, launch thread done after %.2f ms/step" % (a.steps, 1e3 * t_all / a.steps, 1e3 * t_host / a.steps))
    print("%-20s %12s %22s" % ("phase (ends at)", "main ms", "launch lead ms (min/med)"))
    skip = 2                                                   # the first steps start with the queue empty
    for name in [m[0] for m in marks[1:1 + len(per)]]:
        v, l = per[name][skip:], sorted(lead[name][skip:])
        print("%-20s %12.3f %12.1f / %.1f" % (name, sum(v) / len(v), l[0], l[len(l) // 2]))


if __name__ == "__main__":
    main()
