"""Generate mtl_ssl_amd/conv_plans.json on an MI355X: run one training step (and one inference pass)
of each shipped configuration with careful on-line tuning (MTLSSL_TUNE_RUNS=12, plan table ignored)
and save every measured tile choice. Usage: python tools/tune_plans.py [out.json]"""
import os
import sys

os.environ.setdefault("MTLSSL_TUNE_RUNS", "12")
os.environ["MTLSSL_PLAN_DB"] = "0"
os.environ["MTLSSL_AUTOTUNE"] = "1"                   # off by default: the tuner is this tool's
os.environ["MTLSSL_AUX_STREAM"] = "0"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, ops, synthetic, trainer  # noqa: E402

RUNS = [("frcnn_resnet101_coco_mtl.config", 600, 1024), ("rfcn_resnet101_voc_mtl.config", 600, 1024),
        ("frcnn_mobilenet_v1_voc_mtl.config", 600, 1024), ("frcnn_inception_resnet_v2_coco_mtl.config", 800, 1333)]
if os.environ.get("TUNE_ONLY"):                      # e.g. TUNE_ONLY=0 : config[1] alone
    RUNS = [RUNS[int(i)] for i in os.environ["TUNE_ONLY"].split(",")]
for name, H, W in RUNS:
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", name)).read())
    B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
    model = model_builder.build(cfg.model, True, "cuda", seed=0)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = tr.stage_batch(synthetic.make_batch(B, H, W, K, seed=1234, device="cuda"))
    tr.step(batch)
    torch.cuda.synchronize()
    print(name, "tuned", len(ops._tuned), "problems so far", flush=True)
    del model, tr, batch
    torch.cuda.empty_cache()
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "conv_plans.json")
ops.save_plans(out)
if os.environ.get("TUNE_SHOW_US"):                  # the measured table of every problem slower than this
    lim = float(os.environ["TUNE_SHOW_US"]) * 1e-3 * int(os.environ["MTLSSL_TUNE_RUNS"])
    for key, v in sorted(ops._tuned.items()):
        if v is not None and v[2].get(v[0], 0) > lim:
            print(key, "default", v[0], "best", v[1], {c: round(t / int(os.environ["MTLSSL_TUNE_RUNS"]) * 1e3, 1) for c, t in v[2].items()})
changed = sum(1 for v in ops._tuned.values() if v is not None and v[0] != v[1])
print("wrote %s: %d problems, %d with a tile different from the planner's" % (out, len(ops._tuned), changed))
