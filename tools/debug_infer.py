import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
import bench
from mtl_ssl_amd import config, model_builder, synthetic, trainer
from oracle.model import Oracle
text = open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read()
text = text.replace("second_stage_localization_loss_weight",
                    "second_stage_post_processing { batch_non_max_suppression { score_threshold: 0.0 "
                    "iou_threshold: 0.6 max_detections_per_class: 10 max_total_detections: 30 } "
                    "score_converter: SOFTMAX }\n    second_stage_localization_loss_weight", 1)
cfg = config.parse_pipeline_config(text)
tm = model_builder.build(cfg.model, True, "cuda", seed=3)
tr = trainer.Trainer(tm, cfg.train_config, 1)
batch = synthetic.make_batch(2, 160, 224, 5, seed=11, device="cuda", max_gt=4, num_windows=6)
for _ in range(3):
    tr.step(batch)
values = tm.ps.state_dict()
model = model_builder.build(cfg.model, False, "cuda", seed=3, values=values)
x = model.preprocess(batch["images"])
pd = model.predict(x)
pd = model.predict_with_mtl_results(pd)
hp = bench.hyper_params_for_oracle(cfg)
post = dict(score_converter="SOFTMAX", score_threshold=0.0, iou_threshold=0.6, max_detections_per_class=10, max_total_detections=30)
ob, os_, oc, on, aux = Oracle(hp, values).detect(batch["images"].cpu().numpy(), post)
print("num", pd["num_proposals"].cpu().numpy(), aux["num_proposals"])
a = pd["mtl_refined_class_predictions_with_background"].cpu().numpy(); r = aux["class_predictions"]
bad = np.abs(a - r) > 1e-3 * np.abs(r) + 1e-4
print("bad rows", np.unique(np.where(bad)[0]))
pb = pd["proposal_boxes"].cpu().numpy().reshape(-1, 4); rb = aux["proposal_boxes"].reshape(-1, 4)
for row in np.unique(np.where(bad)[0]):
    print(row, pb[row], rb[row], a[row], r[row])
c0 = pd["class_predictions_with_background"].cpu().numpy()
print("cls diff max", np.abs(c0).max())
print("expand win diff")
