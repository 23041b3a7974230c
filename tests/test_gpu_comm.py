"""The RCCL path behind the C ABI (mtlssl_comm_*), run with ONE rank on one GPU: communicator
bootstrap, what RCCL reports, in-place all-reduce / broadcast on a side stream, and the trainer's
overlapped gradient reducer driven through it — side stream, events on the compute streams, buckets
issued from `grad_ready` reports during backward. A one-rank sum is the identity, so the reduced
gradients must equal the un-reduced ones bit for bit while every bucket still goes through
ncclAllReduce. (slim/deployment/model_deploy.py:414-444 is what the collective replaces.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm():
    from mtl_ssl_amd.comm import RcclComm
    c = RcclComm("cuda:0", rank=0, world=1)
    yield c
    c.close()


def test_rccl_reports_the_communicator(comm):
    info = comm.info()
    assert info["backend"] == "rccl" and info["ranks"] == 1 and info["rank"] == 0 and info["device"] == 0
    assert info["rccl_version"] >= 21800          # RCCL tracks NCCL 2.18+ numbering (ROCm 7: 2.2x)


def test_allreduce_and_broadcast_on_a_side_stream(comm):
    side = torch.cuda.Stream()
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda") * 0.5
    ref = x.clone()
    torch.cuda.synchronize()
    for op in ("sum", "max", "min"):
        comm.allreduce(x, op=op, stream=side)
    comm.broadcast(x, 0, stream=side)
    d = torch.tensor([3.25], dtype=torch.float64, device="cuda")
    i = torch.tensor([7, -2], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    comm.allreduce(d, op="max", stream=side)
    comm.allreduce(i, stream=side)
    side.synchronize()
    assert torch.equal(x, ref) and float(d) == 3.25 and i.tolist() == [7, -2]


def test_bad_arguments_are_errors(comm):
    from mtl_ssl_amd.lib import MtlsslError, lib
    with pytest.raises(MtlsslError, match="root"):
        lib().comm_broadcast(comm._h, 1, 16, 5, None)
    with pytest.raises(MtlsslError, match="dtype"):
        lib().comm_allreduce(comm._h, 1, 16, 99, 0, None)


def test_trainer_reduces_through_rccl_with_overlap(comm):
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read())
    batch_args = dict(seed=100, device="cuda", max_gt=4, num_windows=6)
    # reference run: no communicator, no reducer
    m0 = model_builder.build(cfg.model, True, "cuda", seed=3)
    t0 = trainer.Trainer(m0, cfg.train_config, 1)
    assert not t0.reducer.active and m0.ps.grad_ready_hook is None
    b0 = synthetic.make_batch(2, 160, 224, 5, **batch_args)
    t0.forward_backward(b0)
    torch.cuda.synchronize()
    g0 = m0.ps.grads.clone()
    for _ in range(2):
        t0.step(b0)
    # the same through the RCCL reducer
    m1 = model_builder.build(cfg.model, True, "cuda", seed=3)
    t1 = trainer.Trainer(m1, cfg.train_config, 1, comm=comm, reduce_always=True)
    red = t1.reducer
    assert red.active and red.stream is not None and m1.ps.grad_ready_hook is not None
    red.timing = True
    b1 = synthetic.make_batch(2, 160, 224, 5, **batch_args)
    t1.broadcast_weights(0)                      # no-op on one rank, must not disturb anything
    t1.forward_backward(b1)
    early = list(red.launch_order)
    red.finish()
    torch.cuda.synchronize()
    # identity sum. The ROI-crop backward scatters with fp32 atomics, so everything upstream of it (the trunk)
    # is only reproducible to rounding between two runs; the towers and heads downstream are bit-exact.
    ps = m1.ps
    exact = 0
    for sp in ps.trainable_specs:
        a, b = ps._view(ps.grads, sp), g0[sp.offset:sp.offset + sp.size].view(sp.shape)
        if sp.name.startswith(("SecondStage", "MTLClassRefiner")):
            assert torch.equal(a, b), sp.name
            exact += 1
        else:
            scale = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 1e-4 * scale, sp.name
    assert exact > 10
    # every bucket was issued from a grad_ready report during backward, tower/head buckets first
    assert sorted(early) == list(range(len(red.buckets)))
    assert len(red.buckets) == 1 or early[0] > early[-1]
    for _ in range(2):
        t1.step(b1)
    torch.cuda.synchronize()
    assert float((m1.ps.weights - m0.ps.weights).abs().max()) <= 1e-4 * float(m0.ps.weights.abs().max())
    s = red.timing_summary(3)
    assert s["bytes_per_step"] == m1.ps.n_train * 4 and s["allreduce_ms_per_step"] > 0
    assert s["exposed_ms_per_step"] >= 0 and s["hidden_ms_per_step"] >= 0


def test_per_bucket_update_is_bit_identical_to_the_monolithic_update(comm, monkeypatch):
    """With a communicator the clip + momentum update runs bucket by bucket, each behind its own bucket's all-reduce
    (Trainer._update_bucket / GradientReducer.finish_by_bucket) instead of once over all variables after the last
    all-reduce. Per-variable clip_by_norm (slim/learning.py:282-301) makes the update separable per variable: three steps
    of both forms from the same state give the same weights, momentum and shadow weights BIT FOR BIT; the buckets are
    updated in the order their all-reduces were issued, every bucket once; the gradient buffer is left zeroed."""
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read())
    batch = synthetic.make_batch(2, 160, 224, 5, seed=100, device="cuda", max_gt=4, num_windows=6)
    state = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MTLSSL_BUCKET_UPDATE", mode)
        m = model_builder.build(cfg.model, True, "cuda", seed=3)
        t = trainer.Trainer(m, cfg.train_config, 1, comm=comm, reduce_always=True)
        t.reducer = trainer.GradientReducer(m.ps, comm, bucket_bytes=1 << 20, always=True)     # many buckets
        assert t.bucket_update == (mode == "1") and len(t.reducer.buckets) > 4
        orders = []
        for _ in range(3):
            t.step(batch)
            orders.append((list(t.update_order), list(t.reducer.launch_order)))
        torch.cuda.synchronize()
        assert float(m.ps.grads.abs().max()) == 0.0 and m.ps.grads_clean
        state[mode] = (m.ps.weights.clone(), m.ps.accum.clone(), m.ps.eff.clone(), orders, len(t.reducer.buckets))
    wb, ab, eb, orders, nb = state["1"]
    wm, am, em, _, _ = state["0"]
    assert torch.equal(wb, wm) and torch.equal(ab, am) and torch.equal(eb, em)
    for upd, issued in orders:
        assert upd == issued and sorted(upd) == list(range(nb))
        assert upd[0] > upd[-1]                       # tower / head buckets (high offsets) first, the trunk's last


def test_gradient_multipliers_and_frozen_variables_in_the_fused_update():
    """object_detection/trainer.py:389-410 as per-variable entries of the optimizer launch."""
    from mtl_ssl_amd import ops
    from mtl_ssl_amd.params import ParamStore
    ps = ParamStore()
    shapes = {"a/weights": (300, 7), "a/biases": (16,), "b/weights": (40, 40), "b/biases": (8,)}
    for n, sh in shapes.items():
        ps.add(n, sh, ("truncated_normal", 0.5), weight_decay=0.01 if n.endswith("weights") else 0.0)
    ps.finalize("cuda", seed=1)
    rng = np.random.RandomState(0)
    g = rng.randn(ps.n_train).astype(np.float32) * 3
    a = rng.randn(ps.n_train).astype(np.float32)
    ps.grads.copy_(torch.from_numpy(g)); ps.accum.copy_(torch.from_numpy(a))
    w = ps.weights.cpu().numpy().copy()
    mult = {"a/weights": 0.5, "a/biases": 1.0, "b/weights": -1.0, "b/biases": 2.0}   # b/weights frozen
    wd = torch.tensor([s.weight_decay for s in ps.trainable_specs], device="cuda")
    mt = torch.tensor([mult[s.name] for s in ps.trainable_specs], device="cuda")
    lr, mom, clip = 0.1, 0.9, 10.0
    ops.sgd_momentum_clip(ps.weights, ps.grads, ps.accum, ps.var_offsets, ps.max_var_size, lr, mom, clip, 1.0, wd, mt)
    torch.cuda.synchronize()
    got_w, got_a = ps.weights.cpu().numpy(), ps.accum.cpu().numpy()
    offs = ps.var_offsets.cpu().numpy()
    for i, sp in enumerate(ps.trainable_specs):
        sl = slice(offs[i], offs[i + 1])
        if mult[sp.name] < 0:
            np.testing.assert_array_equal(got_w[sl], w[sl]); np.testing.assert_array_equal(got_a[sl], a[sl])
            continue
        gg = (g[sl] + sp.weight_decay * w[sl]) * np.float32(mult[sp.name])
        nrm = np.sqrt((gg.astype(np.float64) ** 2).sum())
        gg = gg * np.float32(min(1.0, clip / nrm))
        aa = mom * a[sl] + gg
        np.testing.assert_allclose(got_a[sl], aa, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(got_w[sl], w[sl] - lr * aa, rtol=2e-5, atol=1e-6)
