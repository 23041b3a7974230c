"""Would running the two images of a batch as two concurrent chains shorten the trunk forward? (un-profiled HIP-event timing)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import config, model_builder, synthetic
cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
fe = model._feature_extractor
x = model.preprocess(synthetic.make_batch(2, 600, 1024, 90, seed=1, device="cuda")["images"])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def full():
    return fe.extract_proposal_features(x, save=False)[0]

def split():
    cur = torch.cuda.current_stream()
    outs = []
    for s, sl in ((s1, x[0:1]), (s2, x[1:2])):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(fe.extract_proposal_features(sl, save=False)[0])
    cur.wait_stream(s1); cur.wait_stream(s2)
    return outs

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

print("trunk forward, batch 2 on one stream : %.3f ms" % timeit(full))
print("trunk forward, 2 x batch 1 on two streams: %.3f ms" % timeit(split))
print("trunk forward, batch 2 on one stream : %.3f ms" % timeit(full))
a = full(); b = split()
print("max diff", float((a - torch.cat(b)).abs().max()))
