"""How the step's three HIP streams share the chip: stream priorities and CU masks, measured on config[1].

The trunk backward (a chain of small GEMMs on the step's stream) runs next to the auxiliary towers' backward (large
tiles, second stream) and the trunk's filter gradients (third stream). A kernel trace shows 12-block launches of the
chain waiting 300 us for a slot while a fresh large-tile launch holds every CU. Two hardware knobs could change who
waits: stream priority (hipStreamCreateWithPriority) and a CU mask on the side streams that leaves a few CUs per XCD
to the chain (hipExtStreamCreateWithCUMask). This tool times the step under each.

    python tools/stream_share_probe.py [steps]
"""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, nn, synthetic, trainer  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(free_per_xcd):
    """A stream whose kernels may not run on the last `free_per_xcd` CUs of each of the 8 XCDs (mask bit i = CU i / 8
    of XCD i % 8)."""
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    keep = n_cu - 8 * free_per_xcd
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(keep):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def run(name, steps, main_priority=None, side_priority=None, free_per_xcd=0):
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read())
    B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
    model = model_builder.build(cfg.model, True, "cuda", seed=0)
    if free_per_xcd:
        model._aux_stream_obj = masked_stream(free_per_xcd)
        model._wgrad_stream_obj = nn.WgradStream(masked_stream(free_per_xcd), group=False)
    elif side_priority is not None:
        model._aux_stream_obj = torch.cuda.Stream(priority=side_priority)
        model._wgrad_stream_obj = nn.WgradStream(torch.cuda.Stream(priority=side_priority), group=False)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    ring = [tr.stage_batch(synthetic.make_batch(B, 600, 1024, K, seed=1234 + i, device="cuda")) for i in range(8)]
    main = torch.cuda.Stream(priority=main_priority) if main_priority is not None else torch.cuda.current_stream()
    with torch.cuda.stream(main):
        for i in range(10):
            tr.step(ring[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(ring[(10 + i) % 8])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    pr = (main.priority, model._aux_stream_obj.priority) if hasattr(main, "priority") else None
    print(json.dumps({"case": name, "ms_per_step": round(ms, 2), "priorities_main_side": pr}), flush=True)
    del tr, model, ring
    torch.cuda.empty_cache()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    print("priority range (least, greatest):", lo, hi, flush=True)
    for rep in range(2):
        run("baseline", steps)
        run("main high priority, side default", steps, main_priority=-1)
        run("main high priority, side low", steps, main_priority=-1, side_priority=1)
        run("main default, side low", steps, side_priority=1)
        for free in (1, 2, 4, 8):
            run("side streams masked off %d CUs per XCD" % free, steps, free_per_xcd=free)
