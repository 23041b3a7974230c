"""Training launcher with the flags of object_detection/train.py:65-98 — what a user of the reference runs:

    python -m mtl_ssl_amd.train --train_dir=/runs/a --pipeline_config_path=configs/frcnn_resnet101_coco_mtl.config
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mtl_ssl_amd.train --train_dir=... ...

One process per GPU; data parallelism comes from the launcher's WORLD_SIZE (RCCL all-reduce of the gradient buckets,
trainer.GradientReducer), not from --num_clones / parameter servers: those flags are accepted so existing command
lines keep working, and a value that asks for the reference's in-process clones is refused with the torchrun line
that does the same job."""
import argparse
import glob
import os
import sys

import numpy as np


def _flags(argv):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--master", default="")
    ap.add_argument("--task", type=int, default=0)
    ap.add_argument("--num_clones", type=int, default=1)
    ap.add_argument("--clone_on_cpu", default="false")
    ap.add_argument("--worker_replicas", type=int, default=1)
    ap.add_argument("--ps_tasks", type=int, default=0)
    ap.add_argument("--train_dir", default="")
    ap.add_argument("--train_tag", default="")
    ap.add_argument("--pipeline_config_path", default="")
    ap.add_argument("--train_config_path", default="")
    ap.add_argument("--input_config_path", default="")
    ap.add_argument("--model_config_path", default="")
    ap.add_argument("--logtostderr", action="store_true")
    ap.add_argument("--num_steps", type=int, default=None, help="overrides train_config.num_steps")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args(argv)


def read_configs(f):
    """train.py:101-155: one TrainEvalPipelineConfig, or the three separate files."""
    from . import config
    if f.pipeline_config_path:
        return config.get_configs_from_pipeline_file(f.pipeline_config_path)
    if not (f.train_config_path and f.input_config_path and f.model_config_path):
        raise SystemExit("give --pipeline_config_path, or all of --model_config_path --train_config_path --input_config_path")
    parts = []
    for field, path in (("model", f.model_config_path), ("train_config", f.train_config_path),
                        ("train_input_reader", f.input_config_path)):
        parts.append("%s {\n%s\n}" % (field, open(path).read()))
    cfg = config.parse_pipeline_config("\n".join(parts))
    return cfg.model, cfg.train_config, cfg.train_input_reader


def record_paths(input_config):
    """input_reader_builder.py:34-65: tf_record_input_reader.input_path (repeated, glob patterns allowed)."""
    paths = []
    reader = input_config.get("tf_record_input_reader")
    if reader is None:
        raise ValueError("the configuration has no train_input_reader { tf_record_input_reader { input_path: ... } }")
    for pat in reader.input_path:
        hits = sorted(glob.glob(pat))
        if not hits:
            raise FileNotFoundError("train_input_reader: nothing matches %r" % pat)
        paths += hits
    if not paths:
        raise ValueError("train_input_reader.tf_record_input_reader.input_path is empty")
    return paths


def main(argv=None):
    f = _flags(sys.argv[1:] if argv is None else argv)
    if f.num_clones != 1 or f.worker_replicas != 1 or f.ps_tasks != 0:
        raise SystemExit("clones / parameter servers are replaced by one process per GPU:\n  python -m torch.distributed.run "
                         "--nproc-per-node %d --master-addr 127.0.0.1 -m mtl_ssl_amd.train --train_dir=%s ..."
                         % (max(f.num_clones * f.worker_replicas, 2), f.train_dir or "DIR"))
    if not f.train_dir:
        raise SystemExit("--train_dir is required")
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from . import input_reader, model_builder, trainer
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        ge.build()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo")            # bootstrap + barriers; gradients travel over RCCL (mtl_ssl_amd.comm)
        dist.barrier()
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dev = torch.device("cuda", torch.cuda.current_device())
    model_config, train_config, input_config = read_configs(f)
    K = int(model_config.faster_rcnn.num_classes)
    B = int(train_config.batch_size)
    if world > 1:                                    # train_config.batch_size is the GLOBAL batch: every clone of the
        if B % world:                                # reference takes batch_size // num_clones images (trainer.py:270)
            raise SystemExit("train_config.batch_size %d does not divide over %d ranks" % (B, world))
        B //= world
    probe = model_builder.build(model_config, True, dev, seed=f.seed)
    rz = model_config.faster_rcnn.image_resizer
    stream = input_reader.batches(record_paths(input_config), K, B, train_config.data_augmentation_options,
                                  np.random.RandomState(f.seed + rank), loop=True, rank=rank, world=world,
                                  # protos/input_reader.proto: shuffle (default true) draws from a queue that holds
                                  # at least min_after_dequeue (default 1000) serialized records
                                  shuffle_buffer=int(input_config.get("min_after_dequeue", 1000) or 0)
                                  if input_config.get("shuffle", True) else 0,
                                  resized_shape=lambda h, w: probe.resized_shape(h, w, rz),
                                  max_pending=64 if world == 1 else 256)

    def next_batch():
        b = next(stream)
        b["images"] = b["images"].to(dev, non_blocking=True)
        return b
    os.makedirs(f.train_dir, exist_ok=True)
    if rank == 0 and f.pipeline_config_path:          # train.py:235-247 keeps the configuration beside the checkpoints
        with open(os.path.join(f.train_dir, "pipeline.config"), "w") as out:
            out.write(open(f.pipeline_config_path).read())
    trainer.train(next_batch, lambda: probe, train_config, master=f.master, task=rank, num_clones=1,
                  worker_replicas=world, is_chief=rank == 0, train_dir=f.train_dir, model_config=model_config,
                  num_steps=f.num_steps)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
