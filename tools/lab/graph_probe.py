"""What would a captured step buy? One training step (forward + losses + backward + update) of a configuration is
captured into a HIP graph on a fixed staged batch and replayed; replay time against the eager step. TIMING PROBE ONLY:
the per-step scalars (sampler stream = f(step), learning rate) are kernel arguments and are frozen into the graph, so a
replay repeats step 0's random draws — the product path does not do this.

    python tools/lab/graph_probe.py [config] [H] [W] -> gpurun_out/r05_graph_probe.txt"""
import os
import sys
import time

os.environ.setdefault("MTLSSL_CHECK_GRADS_CLEAN", "0")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "frcnn_mobilenet_v1_voc_mtl.config"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 600
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
lines = []


def run(aux):
    os.environ["MTLSSL_AUX_STREAM"] = aux
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", name)).read())
    B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
    model = model_builder.build(cfg.model, True, "cuda", seed=0)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    tr.max_steps_in_flight = 0
    batch = tr.stage_batch(synthetic.make_batch(B, H, W, K, seed=1234, device="cuda"))
    for _ in range(6):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        tr.step(batch)
    torch.cuda.synchronize()
    eager = 1e3 * (time.perf_counter() - t0) / 30
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            tr.step(batch)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            tr.step(batch)
    except Exception as e:
        lines.append("%s aux_stream=%s: eager %.2f ms/step; capture failed: %s" % (name, aux, eager, str(e).split("\n")[0][:160]))
        return
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        graph.replay()
    torch.cuda.synchronize()
    rep = 1e3 * (time.perf_counter() - t0) / 30
    lines.append("%s aux_stream=%s: eager %.2f ms/step, captured graph replay %.2f ms/step (frozen per-step scalars: timing only)"
                 % (name, aux, eager, rep))


for aux in (os.environ.get("PROBE_AUX", "0"),):
    try:
        run(aux)
    except Exception as e:
        lines.append("%s aux_stream=%s: failed: %r" % (name, aux, e))
    torch.cuda.synchronize()
out = "\n".join(lines)
print(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r05_graph_probe.txt"), "a").write(out + "\n")
