"""Training driver — object_detection/trainer.py:157-214 (_create_losses), :379-427 (gradient
pipeline) and slim/deployment/model_deploy.py:265-307 (clone loss scaling + gradient sum),
re-designed as one process per GPU: weights replicated in HBM, gradients summed across ranks by
RCCL all-reduce of contiguous buckets of the flat gradient buffer on a side stream, then an
identical per-variable clip + momentum update on every rank.
"""
import os
import time

import torch

from . import ops


def manual_stepping(global_step, boundaries, rates):
    """utils/learning_schedules.py:62-103: rates[#boundaries <= step]."""
    idx = sum(1 for b in boundaries if global_step >= b)
    return rates[idx]


def exponential_decay(global_step, initial, decay_steps, decay_factor, staircase=True):
    """tf.train.exponential_decay (builders/optimizer_builder.py:94-101): initial * factor ** (step / steps), the
    exponent floored when `staircase`."""
    e = global_step / float(decay_steps)
    return initial * decay_factor ** (float(int(e)) if staircase else e)


def optimizer_from_config(optimizer_cfg):
    """builders/optimizer_builder.py:24-118 -> dict(kind, lr_fn, + the optimizer's hyper-parameters).
    kind: 'momentum' (momentum), 'rms_prop' (decay, momentum, epsilon), 'adam' (beta1, beta2, epsilon)."""
    which = optimizer_cfg.which_oneof(["momentum_optimizer", "rms_prop_optimizer", "adam_optimizer"])
    if which is None:
        raise ValueError("Optimizer None not supported.")
    oc = optimizer_cfg[which]
    lr = oc.learning_rate
    kind = lr.which_oneof(["manual_step_learning_rate", "constant_learning_rate", "exponential_decay_learning_rate"])
    if kind == "constant_learning_rate":
        v = float(lr.constant_learning_rate.learning_rate)
        lr_fn = lambda step: v
    elif kind == "manual_step_learning_rate":
        ms = lr.manual_step_learning_rate
        if not ms.schedule:
            raise ValueError("Empty learning rate schedule.")
        bounds = [int(s.step) for s in ms.schedule]
        rates = [float(ms.initial_learning_rate)] + [float(s.learning_rate) for s in ms.schedule]
        lr_fn = lambda step: manual_stepping(step, bounds, rates)
    elif kind == "exponential_decay_learning_rate":
        ed = lr.exponential_decay_learning_rate
        args = (float(ed.initial_learning_rate), int(ed.decay_steps), float(ed.decay_factor), bool(ed.staircase))
        lr_fn = lambda step: exponential_decay(step, *args)
    else:
        raise ValueError("Learning_rate %s not supported." % kind)
    if which == "momentum_optimizer":
        return dict(kind="momentum", lr_fn=lr_fn, momentum=float(oc.momentum_optimizer_value))
    if which == "rms_prop_optimizer":
        return dict(kind="rms_prop", lr_fn=lr_fn, momentum=float(oc.momentum_optimizer_value), decay=float(oc.decay),
                    epsilon=float(oc.epsilon))
    return dict(kind="adam", lr_fn=lr_fn, beta1=float(oc.beta1), beta2=float(oc.beta2), epsilon=float(oc.epsilon))


def learning_rate_fn(optimizer_cfg):
    """(lr_fn, momentum) of a momentum optimizer — kept for callers of the round-1 interface."""
    o = optimizer_from_config(optimizer_cfg)
    return o["lr_fn"], o.get("momentum", 0.0)


class GradientReducer:
    """Cross-replica gradient sum (replaces tf.add_n on the CPU, model_deploy.py:414-444),
    overlapped with backward: buckets are contiguous, variable-aligned slices of the flat gradient
    buffer; every layer reports `ParamStore.grad_ready(var)` when it has enqueued the kernels that
    finalise a variable's gradient, and a bucket's all-reduce (sum) is issued on a side stream as
    soon as its last variable has reported — behind an event on the compute stream(s), so it runs
    while the rest of backward is still executing. xGMI is point-to-point, so buckets are large
    (default 32 MiB) to stay per-link bandwidth-bound rather than latency-bound.

    `comm` is a communicator of mtl_ssl_amd.comm (RcclComm = mtlssl_comm_* = RCCL, the product path;
    GlooComm for CPU tests); None or a single rank disables the reducer unless `always` is set, which
    runs the full machinery — side stream, events, buckets, RCCL calls — on one rank (the 1-GPU test of
    the RCCL path)."""

    def __init__(self, ps, comm=None, bucket_bytes=32 << 20, always=False):
        self.ps, self.comm = ps, comm
        self.world = comm.world if comm is not None else 1
        self.active = comm is not None and (self.world > 1 or always)
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
        self.stream = None
        if self.active and ps.device.type == "cuda":
            from .comm import comm_stream_priority
            self.stream = torch.cuda.Stream(priority=comm_stream_priority())
        self.compute_streams = []          # extra compute streams whose work a bucket may depend on
        self.compute_streams_fn = None     # asked at every bucket launch (a model may create streams lazily)
        self.main_stream = None
        self.pending, self.done, self.seen = [], [], set()
        self.launch_order = []             # bucket ids in the order they were issued (diagnostics/tests)
        self.timing = False                # bench: HIP events around every bucket and around the final wait
        self._ev_buckets, self._ev_wait = [], []
        self._bucket_events = {}
        if self.active:
            ps.grad_ready_hook = self.mark_ready

    def begin_step(self):
        self.main_stream = torch.cuda.current_stream() if self.stream is not None else None
        self.pending = list(self.nvars)
        self.done = [False] * len(self.buckets)
        self.seen = set()
        self.launch_order = []
        self._bucket_events = {}

    def mark_ready(self, spec):
        if not self.active or not self.pending or spec.name in self.seen:
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
        if self.stream is None:                      # CPU tensors (gloo unit tests)
            self.comm.allreduce(g[s:e])
            return
        # the bucket's variables may have been produced on any compute stream (main or auxiliary)
        extra = list(self.compute_streams)
        if self.compute_streams_fn is not None:
            extra += list(self.compute_streams_fn())
        streams = {id(cs): cs for cs in [self.main_stream, torch.cuda.current_stream()] + extra if cs is not None}
        for cs in streams.values():
            ev = torch.cuda.Event()
            ev.record(cs)
            self.stream.wait_event(ev)
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
        self.comm.allreduce(g[s:e], stream=self.stream)
        done = torch.cuda.Event()                    # this bucket's sum is in HBM once the event has passed
        done.record(self.stream)
        self._bucket_events[b] = done
        if self.timing:
            e1.record(self.stream)
            self._ev_buckets.append((e0, e1, (e - s) * 4))

    def finish(self):
        """Issue whatever has not been reduced yet and make the compute stream wait for all of it."""
        if not self.active:
            return
        if not self.done:
            self.begin_step()
        for b in range(len(self.buckets)):
            if not self.done[b]:
                self._launch(b)
        if self.stream is not None:
            cur = torch.cuda.current_stream()
            if self.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
            ops.wait_on(self.stream, "all-reduce of the gradient buckets", cur)
            if self.timing:
                e1.record(cur)
                self._ev_wait.append((e0, e1))
        self.pending = []

    def finish_by_bucket(self):
        """finish() one bucket at a time: issues whatever has not been reduced yet, then yields the bucket ids in the
        order their all-reduces were issued, each AFTER making the compute stream wait for that bucket's all-reduce only —
        so the caller can clip + update bucket b while the all-reduces of the later buckets are still on the wire, and
        only the last bucket's reduce and its slice of the update are exposed at the end of the step (the clone-gradient
        sum followed by apply_gradients, slim/deployment/model_deploy.py:414-444 and slim/learning.py:282-301, per
        bucket: the per-variable clip makes the update separable per variable)."""
        if not self.active:
            return
        if not self.done:
            self.begin_step()
        for b in range(len(self.buckets)):
            if not self.done[b]:
                self._launch(b)
        cur = torch.cuda.current_stream() if self.stream is not None else None
        for n, b in enumerate(list(self.launch_order)):
            if cur is not None:
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                ops.wait_on(self._bucket_events[b], "all-reduce of gradient bucket (last issued)" if n + 1 == len(self.launch_order)
                            else "all-reduce of gradient bucket (earlier ones)", cur)
                if self.timing:
                    e1.record(cur)
                    self._ev_wait.append((e0, e1))
            yield b
        self.pending = []

    def all_reduce(self):
        """Non-overlapped form: reduce every bucket now."""
        self.begin_step()
        self.finish()

    def timing_summary(self, steps):
        """After a synchronize: per-step milliseconds the side stream spent in all-reduce calls, and how
        much of that the compute stream had to wait for (exposed) vs ran under backward (hidden)."""
        busy = sum(a.elapsed_time(b) for a, b, _ in self._ev_buckets)
        nbytes = sum(n for _, _, n in self._ev_buckets)
        exposed = sum(a.elapsed_time(b) for a, b in self._ev_wait)
        steps = max(steps, 1)
        return {"allreduce_ms_per_step": busy / steps, "exposed_ms_per_step": exposed / steps,
                "hidden_ms_per_step": max(busy - exposed, 0.0) / steps, "bytes_per_step": nbytes // steps,
                "buckets": len(self.buckets)}


def bucket_update_tables(ps, buckets):
    """Per bucket (start, end) of the flat gradient buffer: (first variable, one past the last variable, offsets of those
    variables RELATIVE to the bucket start + the bucket length [host list], largest variable extent) — what the fused
    clip + momentum launch needs to run on the bucket's slice alone. Buckets are variable-aligned and consecutive, so the
    variable ranges partition the trainable variables in order."""
    offs = [sp.offset for sp in ps.trainable_specs] + [ps.n_train]
    tabs, v = [], 0
    for s, e in buckets:
        assert offs[v] == s, (offs[v], s)
        v0 = v
        while v < len(offs) - 1 and offs[v] < e:
            v += 1
        rel = [o - s for o in offs[v0:v]] + [e - s]
        tabs.append((v0, v, rel, max(b - a for a, b in zip(rel[:-1], rel[1:]))))
    assert v == len(offs) - 1
    return tabs


def filter_variable_names(names, filter_regex_list, invert=False):
    """utils/variables_helper.py:27-54 on names: drops the names matching (`re.match`) any non-empty pattern of the
    list and returns the rest; with `invert` the complement (the matching names). Empty patterns are ignored."""
    import re
    patterns = [str(r) for r in (filter_regex_list or []) if r]
    kept = []
    for n in names:
        hit = any(re.match(r, n) for r in patterns)
        if (not hit) != bool(invert):
            kept.append(n)
    return kept


def gradient_multipliers(ps, train_config):
    """Per-variable table for the fused optimizer launch, object_detection/trainer.py:389-410:
    grad_multiplier / divide_grad_by_batch on every gradient, bias_grad_multiplier on '.*/biases'
    (utils/variables_helper.py:58-78), freeze_variables (regex list, `re.match`, :29-55,100-118) as a
    negative entry = the variable is left out of the update. None when every entry is 1."""
    base = float(train_config.grad_multiplier) if train_config.grad_multiplier else 1.0
    if train_config.divide_grad_by_batch:
        base /= float(train_config.batch_size)
    bias = float(train_config.bias_grad_multiplier) if train_config.bias_grad_multiplier else None
    names = [sp.name for sp in ps.trainable_specs]
    biases = set(filter_variable_names(names, [".*/biases"], invert=True)) if bias is not None else ()
    frozen = set(filter_variable_names(names, list(train_config.freeze_variables or []), invert=True))
    mult = []
    for n in names:
        m = base
        if n in biases:
            m *= bias
        if n in frozen:
            m = -1.0
        mult.append(m)
    if all(m == 1.0 for m in mult):
        return None
    return torch.tensor(mult, dtype=torch.float32, device=ps.device)


class Trainer:
    """One training replica. step(batch) = forward + loss + backward + all-reduce + update."""

    def __init__(self, model, train_config, world_size=1, comm=None, reduce_always=False):
        self.model, self.ps, self.cfg = model, model.ps, train_config
        self.opt = optimizer_from_config(train_config.optimizer)
        self.lr_fn, self.momentum = self.opt["lr_fn"], self.opt.get("momentum", 0.0)
        # second optimizer slot (RMSProp: momentum next to the mean square in ps.accum, which TensorFlow starts at
        # one; Adam: v next to m)
        self.slot1 = None
        if self.opt["kind"] != "momentum":
            self.slot1 = torch.zeros_like(self.ps.accum)
            if self.opt["kind"] == "rms_prop":
                self.ps.accum.fill_(1.0)
        self.clip = float(train_config.gradient_clipping_by_norm)
        self.world = world_size
        self.global_step = 0
        if comm is None and world_size > 1:
            from . import comm as comm_mod
            comm = comm_mod.default_comm(self.ps.device)
        if comm is not None and comm.world != world_size:
            raise ValueError("communicator has %d ranks, trainer was given world_size %d" % (comm.world, world_size))
        self.comm = comm
        self.reducer = GradientReducer(self.ps, comm, always=reduce_always)
        wd = [s.weight_decay for s in self.ps.trainable_specs]
        self.var_wd = (torch.tensor(wd, dtype=torch.float32, device=self.ps.device)
                       if any(w != 0.0 for w in wd) else None)
        self.var_mult = gradient_multipliers(self.ps, train_config)
        import os
        self.split_loss = os.environ.get("MTLSSL_SPLIT_LOSS", "1") != "0" and self.ps.device.type == "cuda"
        # the momentum update is the gradient buffer's last reader of a step: it leaves zeros behind, which saves the
        # next step's memset launch over every parameter (the first step starts from ParamStore's zero-initialised buffer)
        self.zero_in_update = os.environ.get("MTLSSL_ZERO_IN_UPDATE", "1") != "0" and self.ps.device.type == "cuda"
        self.max_steps_in_flight = int(os.environ.get("MTLSSL_MAX_STEPS_IN_FLIGHT", "2"))     # 0: unbounded (rounds 1-4)
        self._step_events = []
        # momentum update and shadow-weight fold in one launch: only when no SCALE vector trains — frozen BatchNorm, or a
        # normaliser without gamma whose beta alone trains (Inception-ResNet-v2's slim.batch_norm(scale=False): the folded
        # scale is 1/sqrt(var + eps), a constant; beta only moves the shift, which bn_refresh recomputes)
        self.fuse_fold = (os.environ.get("MTLSSL_FUSE_FOLD", "1") != "0"
                          and not any(getattr(l, "bn_trainable", False) and getattr(l, "gamma", None) is not None
                                      for l in model.layers))
        # with a communicator the clip + momentum update runs per gradient bucket, right behind that bucket's all-reduce
        # (GradientReducer.finish_by_bucket): the step no longer ends with "last all-reduce -> norms over all 78 M
        # parameters -> update over all of them" in series. Same arithmetic per variable: bit-identical weights.
        self.bucket_update = (os.environ.get("MTLSSL_BUCKET_UPDATE", "1") != "0" and self.reducer.active
                              and self.opt["kind"] == "momentum")
        self._bucket_tabs = None
        self.update_order = []             # bucket ids in the order their slices were updated (diagnostics / tests)
        own = os.environ.get("MTLSSL_STEP_STREAM", "auto")
        use = (self.reducer.active if own == "auto" else own == "1") and self.ps.device.type == "cuda"
        # MTLSSL_STEP_STREAM_PRIORITY: HIP priority of the step's own (main) stream when it has one (-1 = above the side streams)
        self.step_stream = (torch.cuda.Stream(device=self.ps.device, priority=int(os.environ.get("MTLSSL_STEP_STREAM_PRIORITY", "0")))
                            if use else None)
        # builders/optimizer_builder.py:105-111: tf.contrib.opt.MovingAverageOptimizer keeps an exponential
        # moving average of every variable beside it (the trainer's plain Saver stores both); decay as given.
        self.ema = None
        if train_config.optimizer.use_moving_average:
            self.ema_decay = float(train_config.optimizer.moving_average_decay)
            self.ema = self.ps.weights.clone()

    def broadcast_weights(self, root=0):
        """C2: identical initial values on every replica — the reference gets this from its
        single CPU copy of each variable (model_deploy.py:640-675)."""
        if self.comm is None or self.comm.world == 1:
            return
        self.comm.broadcast(self.ps.weights, root)
        self.comm.broadcast(self.ps.frozen, root)
        self.comm.broadcast(self.ps.accum, root)
        if self.slot1 is not None:
            self.comm.broadcast(self.slot1, root)
        if self.ps.device.type == "cuda":
            torch.cuda.current_stream().synchronize()
        self.model.prepare()

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
        if not self.ps.grads_clean:          # the momentum update leaves zeros behind (MTLSSL_ZERO_IN_UPDATE, default on)
            self.ps.grads.zero_()
        elif os.environ.get("MTLSSL_CHECK_GRADS_CLEAN") == "1":      # debug / test suite: the skipped memset was safe
            assert float(self.ps.grads.abs().max().item()) == 0.0, \
                "gradient buffer written since the last zeroing update without ParamStore.mark_grads_dirty()"
        self.ps.grads_clean = False
        self.reducer.compute_streams = m.compute_streams()
        self.reducer.compute_streams_fn = m.compute_streams
        self.reducer.begin_step()
        ops.mark("step_start")
        images = m.preprocess(batch["images"])
        pd = m.predict_for_training(images)       # predict + predict_with_window + predict_edgemask
        ops.mark("predict")
        mtl = m._mtl
        side = m._aux_stream() if (mtl.refine and self.split_loss) else None
        if side is not None:
            # every loss term but the refiner's is a chain of small latency-bound kernels (target assignment over
            # 14 453 anchors, samplers, reductions): on the side stream they run under the refiner's tower forward
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                m.loss(pd, loss_scale=1.0 / self.world, part="early")
            pd = m.predict_with_mtl_results(pd)
            ops.wait_on(side, "losses: early part on the aux stream", cur)
            losses = m.loss(pd, loss_scale=1.0 / self.world, part="late")
        else:
            if mtl.refine:
                pd = m.predict_with_mtl_results(pd)
            losses = m.loss(pd, loss_scale=1.0 / self.world)
        ops.mark("refine_and_losses")
        m.backward(pd)
        ops.mark("backward")
        self._pd = pd
        return losses

    def apply_gradients(self):
        """trainer.py:379-427: cross-replica sum, gradient multipliers / frozen variables, per-variable
        clip_by_norm, momentum update."""
        self.reducer.compute_streams = self.model.compute_streams()
        lr = self.lr_fn(self.global_step)
        ps = self.ps
        o = self.opt
        folded = False
        if self.bucket_update:
            folded = self.fuse_fold and ps.device.type == "cuda" and ps.eff is not None
            self.update_order = []
            for b in self.reducer.finish_by_bucket():
                self._update_bucket(b, lr, folded)
                self.update_order.append(b)
            ps.grads_clean = self.zero_in_update
            if self.ema is not None:
                ops.axpby(ps.weights, self.ema, 1.0 - self.ema_decay, self.ema_decay)
            self.model.refold(folded=folded)
            ops.mark("update")
            self.global_step += 1
            return
        self.reducer.finish()
        if o["kind"] == "momentum":
            # with every BatchNorm frozen the scale vectors are constants: the update launch refreshes the shadow
            # weights itself and the separate fold (a second pass over all 78 M parameters) is skipped
            folded = self.fuse_fold and ps.device.type == "cuda" and ps.eff is not None
            ops.sgd_momentum_clip(ps.weights, ps.grads, ps.accum, ps.var_offsets, ps.max_var_size, lr,
                                  self.momentum, self.clip, 1.0, self.var_wd, self.var_mult,
                                  fold=ps if folded else None, zero_grads=self.zero_in_update)
            ps.grads_clean = self.zero_in_update
        elif o["kind"] == "rms_prop":
            ops.adaptive_update_clip(1, ps.weights, ps.grads, ps.accum, self.slot1, ps.var_offsets, ps.max_var_size, lr,
                                     o["decay"], o["momentum"], o["epsilon"], self.clip, 1.0, self.var_wd, self.var_mult)
        else:
            t = self.global_step + 1                        # AdamOptimizer's beta powers after t updates
            lr_t = lr * (1.0 - o["beta2"] ** t) ** 0.5 / (1.0 - o["beta1"] ** t)
            ops.adaptive_update_clip(2, ps.weights, ps.grads, ps.accum, self.slot1, ps.var_offsets, ps.max_var_size, lr_t,
                                     o["beta1"], o["beta2"], o["epsilon"], self.clip, 1.0, self.var_wd, self.var_mult)
        if self.ema is not None:
            ops.axpby(ps.weights, self.ema, 1.0 - self.ema_decay, self.ema_decay)
        self.model.refold(folded=folded)
        ops.mark("update")
        self.global_step += 1

    def _update_bucket(self, b, lr, folded):
        """clip_by_norm + momentum (+ shadow-weight fold, + zeroing of the gradients) on the variables of bucket b only."""
        ps = self.ps
        if self._bucket_tabs is None:
            self._bucket_tabs = [(v0, v1, torch.tensor(rel, dtype=torch.int32, device=ps.device), mx)
                                 for v0, v1, rel, mx in bucket_update_tables(ps, self.reducer.buckets)]
        s, e = self.reducer.buckets[b]
        v0, v1, rel, mx = self._bucket_tabs[b]
        fold = (ps.eff[s:e], ps.fold_ptrs[v0:v1], ps.fold_len[v0:v1]) if folded else None
        ops.sgd_momentum_clip(ps.weights[s:e], ps.grads[s:e], ps.accum[s:e], rel, mx, lr, self.momentum, self.clip, 1.0,
                              None if self.var_wd is None else self.var_wd[v0:v1],
                              None if self.var_mult is None else self.var_mult[v0:v1],
                              fold=fold, zero_grads=self.zero_in_update)

    def step(self, batch):
        """One training step. With a communicator the step runs on a stream of its own instead of the legacy
        default stream: stream 0 synchronises implicitly with every blocking stream of the process (RCCL keeps
        internal ones), which made the main path wait for the auxiliary stream's whole queue in the middle of
        backward (a 10 ms bubble in the kernel trace of the 1-rank RCCL run)."""
        self._throttle()
        if self.step_stream is None:
            losses = self.forward_backward(batch)
            self.apply_gradients()
            self._step_issued()
            return losses
        cur = torch.cuda.current_stream()
        self.step_stream.wait_stream(cur)
        with torch.cuda.stream(self.step_stream):
            losses = self.forward_backward(batch)
            self.apply_gradients()
        cur.wait_stream(self.step_stream)
        self._step_issued()
        return losses

    # The launch thread needs ~25 ms to enqueue a 53-ms step, so left alone it runs as far ahead of the device as the
    # HIP queues allow (measured: 85-300 ms, i.e. up to five steps). Every step in flight pins its temporaries — a block
    # handed to a side stream (record_stream) returns to the caching allocator only when the DEVICE has passed that
    # point — so the allocator's reserve grew to five times the live bytes (52.6 GB for 10.3 GB).
    # Two steps in flight keep the device's queues full (the host is a whole step ahead) and bound the reserve.
    def _throttle(self):
        lim = self.max_steps_in_flight
        if lim <= 0 or self.ps.device.type != "cuda":
            return
        ev = self._step_events
        while len(ev) >= lim:
            ev.pop(0).synchronize()

    def _step_issued(self):
        if self.max_steps_in_flight > 0 and self.ps.device.type == "cuda":
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream())
            self._step_events.append(e)


def collective_verdict(comm, code, device):
    """max over the data-parallel ranks of a small non-negative failure code (0 = fine). Every rank must call it at
    the same point of the step loop; with one rank (or no communicator) it is the identity."""
    if comm is None or comm.world <= 1:
        return int(code)
    flag = torch.tensor([int(code)], dtype=torch.int32, device=device)
    comm.allreduce(flag, op="max")
    return int(flag.item())


def train(create_tensor_dict_fn, create_model_fn, train_config, master="", task=0, num_clones=1,
          worker_replicas=1, clone_on_cpu=False, ps_tasks=0, worker_job_name="lonely_worker",
          is_chief=True, train_dir=None, num_examples=0, total_configs=None, model_config=None,
          is_first_training=True, num_steps=None, log_every=10, save_interval_secs=600):
    """object_detection/trainer.py:217-219 signature. `create_tensor_dict_fn()` yields one batch
    dict per call (see mtl_ssl_amd.synthetic.make_batch for the field contract);
    `create_model_fn()` returns a built FasterRCNNMetaArch. Parameter-server arguments
    (master, ps_tasks, worker_job_name, clone_on_cpu) are accepted and ignored: data parallelism
    here is one process per GPU over RCCL."""
    import os
    import torch.distributed as dist
    from . import checkpoint
    world = dist.get_world_size() if dist.is_initialized() else 1
    model = create_model_fn()
    trainer = Trainer(model, train_config, world)
    # Resume from train_dir if a state file is there (slim.learning.train restores the latest checkpoint of
    # logdir), else initialise from fine_tune_checkpoint (trainer.py:309-356: a Saver over restore_map() that
    # FAILS when the checkpoint cannot be read). Containers: this build's .npz keyed by the reference's variable
    # names, or TensorFlow's own checkpoint files (mtl_ssl_amd/tf_checkpoint.py).
    state = os.path.join(train_dir, "model.ckpt.npz") if train_dir else None
    if state and os.path.exists(state):
        trainer.global_step = checkpoint.load(state, model.ps, trainer)
        model.prepare()
    elif train_config.fine_tune_checkpoint:
        mtl = model_config.mtl if model_config is not None else model._mtl
        ckpt = checkpoint.open_checkpoint(str(train_config.fine_tune_checkpoint))
        done = checkpoint.init_from_checkpoint(model, ckpt, train_config, mtl)
        if not done:
            raise ValueError("fine_tune_checkpoint %s holds none of the %d variables the restore map asks for"
                             % (train_config.fine_tune_checkpoint, len(model.ps.specs)))
        if is_chief:
            print("restored %d of %d variables from %s" % (len(done), len(model.ps.specs),
                                                            train_config.fine_tune_checkpoint))
    trainer.broadcast_weights(0)
    # num_steps == 0 trains indefinitely (train.proto:42-44; slim.learning.train number_of_steps=None)
    steps = num_steps if num_steps is not None else (int(train_config.num_steps) or None)
    save_secs = float(save_interval_secs) if save_interval_secs else 0.0
    last_save = time.time()

    def save_state():
        if state and is_chief:
            os.makedirs(train_dir, exist_ok=True)
            tmp = state + ".tmp.npz"
            checkpoint.save(tmp, model.ps, trainer.global_step, trainer)
            os.replace(tmp, state)             # a crash mid-write never clobbers the previous state

    log = []
    while steps is None or trainer.global_step < steps:
        t0 = time.time()
        losses = trainer.step(create_tensor_dict_fn())
        if trainer.global_step % log_every == 0 or trainer.global_step == steps:
            if model.ps.device.type == "cuda":
                torch.cuda.synchronize()
            total = float(sum(v.item() for v in losses.values()))
            # tf.check_numerics (:207-209) and the device-side invariants, decided COLLECTIVELY: a rank that raised on
            # its own would leave the others blocked in the next step's all-reduce for good, so every rank learns the
            # worst verdict first (one 4-byte max all-reduce on the log steps only) and all of them raise together
            bad, why = 0, None
            if not (total == total and abs(total) != float("inf")):
                bad, why = 1, FloatingPointError("LossTensor is inf or nan")
            elif hasattr(model, "check_device_flags"):
                try:
                    model.check_device_flags()
                except RuntimeError as e:
                    bad, why = 2, e
            worst = collective_verdict(trainer.comm, bad, model.ps.device)
            if worst and not bad:
                why = RuntimeError("another data-parallel rank reported %s; stopping with it"
                                   % ("a non-finite loss" if worst == 1 else "a device-side invariant violation"))
            bad = worst
            if bad:
                if trainer.comm is not None:
                    trainer.comm.close()
                raise why
            dt = time.time() - t0
            log.append({"step": trainer.global_step, "loss": total, "sec_per_step": dt})
            if is_chief:
                print("global step %d: loss = %.4f (%.3f sec/step)" % (trainer.global_step, total, dt))
        if save_secs and time.time() - last_save >= save_secs:          # trainer.py:464-466 save_interval_secs
            save_state()
            last_save = time.time()
    save_state()
    return trainer, log
