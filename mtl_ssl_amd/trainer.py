"""Training driver — object_detection/trainer.py:157-214 (_create_losses), :379-427 (gradient
pipeline) and slim/deployment/model_deploy.py:265-307 (clone loss scaling + gradient sum),
re-designed as one process per GPU: weights replicated in HBM, gradients summed across ranks by
RCCL all-reduce of contiguous buckets of the flat gradient buffer on a side stream, then an
identical per-variable clip + momentum update on every rank.
"""
import time

import torch

from . import ops


def manual_stepping(global_step, boundaries, rates):
    """utils/learning_schedules.py:62-103: rates[#boundaries <= step]."""
    idx = sum(1 for b in boundaries if global_step >= b)
    return rates[idx]


def learning_rate_fn(optimizer_cfg):
    """builders/optimizer_builder.py:24-118 (momentum / manual_step or constant)."""
    which = optimizer_cfg.which_oneof(["momentum_optimizer", "rms_prop_optimizer", "adam_optimizer"])
    if which != "momentum_optimizer":
        raise ValueError("Optimizer %s not supported (the paper configs use momentum_optimizer)." % which)
    mo = optimizer_cfg.momentum_optimizer
    lr = mo.learning_rate
    kind = lr.which_oneof(["manual_step_learning_rate", "constant_learning_rate",
                           "exponential_decay_learning_rate"])
    if kind == "constant_learning_rate":
        v = float(lr.constant_learning_rate.learning_rate)
        return (lambda step: v), float(mo.momentum_optimizer_value)
    if kind == "manual_step_learning_rate":
        ms = lr.manual_step_learning_rate
        if not ms.schedule:
            raise ValueError("Empty learning rate schedule.")
        bounds = [int(s.step) for s in ms.schedule]
        rates = [float(ms.initial_learning_rate)] + [float(s.learning_rate) for s in ms.schedule]
        return (lambda step: manual_stepping(step, bounds, rates)), float(mo.momentum_optimizer_value)
    raise ValueError("Learning_rate %s not supported." % kind)


class GradientReducer:
    """Cross-replica gradient sum (replaces tf.add_n on the CPU, model_deploy.py:414-444),
    overlapped with backward: buckets are contiguous, variable-aligned slices of the flat gradient
    buffer; every layer reports `ParamStore.grad_ready(var)` when it has enqueued the kernels that
    finalise a variable's gradient, and a bucket's all-reduce (sum) is issued on a side stream as
    soon as its last variable has reported — behind an event on the compute stream(s), so it runs
    while the rest of backward is still executing. xGMI is point-to-point, so buckets are large
    (default 32 MiB) to stay per-link bandwidth-bound rather than latency-bound."""

    def __init__(self, ps, bucket_bytes=32 << 20):
        import torch.distributed as dist
        self.dist = dist
        self.ps = ps
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        per = max(bucket_bytes // 4, 1)
        self.buckets, self.var_bucket, self.nvars = [], {}, []
        start, count = 0, 0
        for sp in ps.trainable_specs:                     # variable-aligned, >= `per` floats each
            self.var_bucket[sp.name] = len(self.buckets)
            count += 1
            end = sp.offset + -(-sp.size // 64) * 64
            if end - start >= per:
                self.buckets.append((start, end)); self.nvars.append(count)
                start, count = end, 0
        if count:
            self.buckets.append((start, ps.n_train)); self.nvars.append(count)
        if self.buckets:
            self.buckets[-1] = (self.buckets[-1][0], ps.n_train)
        self.stream = torch.cuda.Stream() if (self.world > 1 and ps.device.type == "cuda") else None
        self.compute_streams = []          # extra compute streams whose work a bucket may depend on
        self.main_stream = None
        self.pending, self.done, self.seen = [], [], set()
        self.launch_order = []             # bucket ids in the order they were issued (diagnostics/tests)
        if self.world > 1:
            ps.grad_ready_hook = self.mark_ready

    def begin_step(self):
        self.main_stream = torch.cuda.current_stream() if self.stream is not None else None
        self.pending = list(self.nvars)
        self.done = [False] * len(self.buckets)
        self.seen = set()
        self.launch_order = []

    def mark_ready(self, spec):
        if self.world == 1 or not self.pending or spec.name in self.seen:
            return
        self.seen.add(spec.name)
        b = self.var_bucket[spec.name]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        s, e = self.buckets[b]
        g = self.ps.grads
        self.done[b] = True
        self.launch_order.append(b)
        if self.stream is None:                      # gloo / CPU path used by the unit tests
            self.dist.all_reduce(g[s:e])
            return
        # the bucket's variables may have been produced on any compute stream (main or auxiliary)
        streams = {id(cs): cs for cs in [self.main_stream, torch.cuda.current_stream()] + self.compute_streams
                   if cs is not None}
        for cs in streams.values():
            ev = torch.cuda.Event()
            ev.record(cs)
            self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            self.dist.all_reduce(g[s:e])

    def finish(self):
        """Issue whatever has not been reduced yet and make the compute stream wait for all of it."""
        if self.world == 1:
            return
        if not self.done:
            self.begin_step()
        for b in range(len(self.buckets)):
            if not self.done[b]:
                self._launch(b)
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.pending = []

    def all_reduce(self):
        """Non-overlapped form: reduce every bucket now."""
        self.begin_step()
        self.finish()


class Trainer:
    """One training replica. step(batch) = forward + loss + backward + all-reduce + update."""

    def __init__(self, model, train_config, world_size=1):
        self.model, self.ps, self.cfg = model, model.ps, train_config
        self.lr_fn, self.momentum = learning_rate_fn(train_config.optimizer)
        self.clip = float(train_config.gradient_clipping_by_norm)
        self.world = world_size
        self.global_step = 0
        self.reducer = GradientReducer(self.ps)
        wd = [s.weight_decay for s in self.ps.trainable_specs]
        self.var_wd = (torch.tensor(wd, dtype=torch.float32, device=self.ps.device)
                       if any(w != 0.0 for w in wd) else None)
        if train_config.optimizer.use_moving_average:
            # the reference wraps the optimizer in an EMA of the weights for evaluation only
            # (builders/optimizer_builder.py:105-111); every paper config disables it.
            pass

    def stage_batch(self, batch):
        """Move a batch's groundtruth to HBM once (padded device tensors); later provide() calls
        re-install the staged tensors without touching the host."""
        self.provide(batch)
        m = self.model
        batch["_staged"] = (dict(m._gt), m._window, m._edgemask)
        return batch

    def provide(self, batch):
        m = self.model
        if "_staged" in batch:
            gt, m._window, m._edgemask = batch["_staged"]
            m._gt = dict(gt)
            return
        m.provide_groundtruth(batch["groundtruth_boxes"], batch["groundtruth_classes"],
                              batch.get("groundtruth_closeness"))
        if m._mtl.window:
            m.provide_window(batch["window_boxes"], batch["window_classes"])
        if m._mtl.edgemask:
            m.provide_edgemask(batch["groundtruth_edgemask"])

    def forward_backward(self, batch):
        """trainer.py:157-214 + backward. Returns the loss dict (device scalars)."""
        m = self.model
        m.step = self.global_step
        self.provide(batch)
        self.ps.grads.zero_()
        self.reducer.compute_streams = m.compute_streams()
        self.reducer.begin_step()
        images = m.preprocess(batch["images"])
        pd = m.predict_for_training(images)       # predict + predict_with_window + predict_edgemask
        mtl = m._mtl
        if mtl.refine:
            pd = m.predict_with_mtl_results(pd)
        losses = m.loss(pd, loss_scale=1.0 / self.world)
        m.backward(pd)
        self._pd = pd
        return losses

    def apply_gradients(self):
        """trainer.py:379-427: cross-replica sum, per-variable clip_by_norm, momentum update."""
        self.reducer.compute_streams = self.model.compute_streams()
        self.reducer.finish()
        lr = self.lr_fn(self.global_step)
        ps = self.ps
        ops.sgd_momentum_clip(ps.weights, ps.grads, ps.accum, ps.var_offsets, ps.max_var_size, lr,
                              self.momentum, self.clip, 1.0, self.var_wd)
        self.model.refold()
        self.global_step += 1

    def step(self, batch):
        losses = self.forward_backward(batch)
        self.apply_gradients()
        return losses


def train(create_tensor_dict_fn, create_model_fn, train_config, master="", task=0, num_clones=1,
          worker_replicas=1, clone_on_cpu=False, ps_tasks=0, worker_job_name="lonely_worker",
          is_chief=True, train_dir=None, num_examples=0, total_configs=None, model_config=None,
          is_first_training=True, num_steps=None, log_every=10):
    """object_detection/trainer.py:217-219 signature. `create_tensor_dict_fn()` yields one batch
    dict per call (see mtl_ssl_amd.synthetic.make_batch for the field contract);
    `create_model_fn()` returns a built FasterRCNNMetaArch. Parameter-server arguments
    (master, ps_tasks, worker_job_name, clone_on_cpu) are accepted and ignored: data parallelism
    here is one process per GPU over RCCL."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    import os
    from . import checkpoint
    model = create_model_fn()
    trainer = Trainer(model, train_config, world)
    # Resume from train_dir if a state file is there, else initialise from fine_tune_checkpoint
    # (trainer.py:309-356; slim.learning.train restores the latest checkpoint of logdir). The
    # container is .npz keyed by the reference's variable names (mtl_ssl_amd/checkpoint.py).
    state = os.path.join(train_dir, "model.ckpt.npz") if train_dir else None
    if state and os.path.exists(state):
        trainer.global_step = checkpoint.load(state, model.ps)
        model.prepare()
    elif train_config.fine_tune_checkpoint and os.path.exists(str(train_config.fine_tune_checkpoint)):
        import numpy as np
        mtl = model_config.mtl if model_config is not None else model._mtl
        checkpoint.init_from_checkpoint(model, np.load(str(train_config.fine_tune_checkpoint)), train_config, mtl)
    steps = num_steps if num_steps is not None else (int(train_config.num_steps) or 10)
    log = []
    for _ in range(max(steps - trainer.global_step, 0)):
        t0 = time.time()
        losses = trainer.step(create_tensor_dict_fn())
        if trainer.global_step % log_every == 0 or trainer.global_step == steps:
            torch.cuda.synchronize()
            total = float(sum(v.item() for v in losses.values()))
            if not (total == total and abs(total) != float("inf")):
                raise FloatingPointError("LossTensor is inf or nan")     # tf.check_numerics, :207-209
            dt = time.time() - t0
            log.append({"step": trainer.global_step, "loss": total, "sec_per_step": dt})
            if is_chief:
                print("global step %d: loss = %.4f (%.3f sec/step)" % (trainer.global_step, total, dt))
    if state and is_chief:
        os.makedirs(train_dir, exist_ok=True)
        checkpoint.save(state, model.ps, trainer.global_step)
    return trainer, log
