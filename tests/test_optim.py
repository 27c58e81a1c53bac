"""Fused clip + AdamW + warm-up cosine step (csrc/optim.hip) vs the reference's torch.optim.AdamW(betas=(0.9, 0.98)) +
clip_grad_norm_(10) + WarmupCosineScheduler (lightning.py:48-52, train.py:41, cosine.py:6-25)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosine import WarmupCosineScheduler  # noqa: E402

from auto_avsr_amd.optim import FusedAdamW  # noqa: E402


@pytest.mark.parametrize("clip", [10.0, 0.5, 0.0])
def test_fused_adamw_matches_torch(dev, clip):
    torch.manual_seed(0)
    shapes = [(7,), (33, 5), (4097,), (64, 65), (3, 1, 5, 7, 7), (1,)]
    ref_p = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    lr, wd, warm, total = 1e-2, 0.03, 3, 10
    ref = torch.optim.AdamW(ref_p, lr=lr, betas=(0.9, 0.98), weight_decay=wd)
    sched = WarmupCosineScheduler(ref, warm, total, 1)
    ours = FusedAdamW(our_p, lr=lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd, max_grad_norm=clip, warmup_steps=warm,
                      total_steps=total)
    for it in range(6):
        gs = [torch.randn(s) * (3.0 if it % 2 else 0.3) for s in shapes]
        for p, q, g in zip(ref_p, our_p, gs):
            p.grad = g.clone()
            q.grad = g.clone().to(dev)  # fresh tensors every step, as after a backward
        lr_used = sched.get_last_lr()[0]  # torch applies the schedule of the previous scheduler.step()
        norm = torch.nn.utils.clip_grad_norm_(ref_p, clip) if clip > 0 else torch.linalg.vector_norm(torch.cat([g.flatten() for g in gs]))
        ref.step()
        sched.step()
        ours.step()
        assert abs(ours.last_grad_norm - float(norm)) < 1e-4 * max(1.0, float(norm))
        assert ours.step_count == it + 1
        # schedule: the reference scheduler is constructed with step count 1 (factor 1/warm), then stepped after each
        # optimizer step; the fused step uses factor(step) with step = 1, 2, ...
        assert abs(ours.last_lr - lr_used) < 1e-9 + 1e-6 * lr_used, (it, ours.last_lr, lr_used)
        for p, q in zip(ref_p, our_p):
            assert (p.detach() - q.detach().cpu()).abs().max() < 2e-6 * max(1.0, float(p.detach().abs().max()))


def test_fused_adamw_cast_weights(dev):
    """cast_weights=True: same parameters / moments as the plain fused step, and the bf16
    operand copies of the registered Linear weights (plain, transposed, slices of a Q/K/V concatenation) equal a fresh
    re-cast of the updated weights -- with refresh_weight_cache() turned into a no-op for them."""
    from auto_avsr_amd import functional as AF

    torch.manual_seed(5)
    shapes = [(128, 64), (64, 128), (64, 64), (64, 64), (64, 64), (72, 40), (64,), (5049,), (3, 1, 5, 7, 7)]

    def make(cast):
        torch.manual_seed(11)
        ps = [torch.nn.Parameter(torch.randn(s).to(dev)) for s in shapes]
        return ps, FusedAdamW(ps, lr=1e-2, weight_decay=0.03, max_grad_norm=1.0, warmup_steps=2, total_steps=8,
                              cast_weights=cast)

    AF.invalidate_weight_cache()
    pa, oa = make(False)
    pb, ob = make(True)
    # register copies of pb's 2-D weights the way the modules do: forward copy, data-gradient (transposed) copy, fused QKV
    copies = [(pb[0], False), (pb[0], True), (pb[1], True), (pb[5], False), (pb[5], True)]
    got = [AF._w_bf16(w, t) for w, t in copies]
    cat = AF._w_bf16_cat([pb[2], pb[3], pb[4]], False)
    catT = AF._w_bf16_cat([pb[2], pb[3], pb[4]], True)
    for it in range(3):
        gs = [torch.randn(s) * (3.0 if it % 2 else 0.3) for s in shapes]
        for p, q, g in zip(pa, pb, gs):
            p.grad, q.grad = g.clone().to(dev), g.clone().to(dev)
        oa.step()
        ob.step()
        for p, q, ma, mb, va, vb in zip(pa, pb, oa.exp_avg, ob.exp_avg, oa.exp_avg_sq, ob.exp_avg_sq):
            # same arithmetic in two kernels: equal up to the compiler's choice of fused multiply-adds
            for a, b in ((p.detach(), q.detach()), (ma, mb), (va, vb)):
                assert (a - b).abs().max() <= 1e-6 * max(1.0, float(a.abs().max()))
        assert AF._wgen["owner"] is not None and AF._wgen["owner"]() is ob
        launches = []
        orig = AF.ops.multi_cast_transpose
        AF.ops.multi_cast_transpose = lambda *a, **k: launches.append(a)
        try:
            AF.refresh_weight_cache()  # nothing left to do
        finally:
            AF.ops.multi_cast_transpose = orig
        assert not launches
        for (w, t), c in zip(copies, got):
            ref = w.detach().cpu().to(torch.bfloat16)
            if t:
                assert torch.equal(c.cpu()[:, :w.shape[0]], ref.t()) and not c.cpu()[:, w.shape[0]:].any()
            else:
                assert torch.equal(c.cpu(), ref)
        full = torch.cat([pb[2], pb[3], pb[4]]).detach().cpu().to(torch.bfloat16)
        assert torch.equal(cat.cpu(), full) and torch.equal(catT.cpu(), full.t())
    # a weight changed behind the optimizer's back (load_state_dict, manual edits: the tensor version moves) is re-cast
    # on its next use even while the claim stands -- also when it is a slice of a concatenation
    with torch.no_grad():
        pb[3].mul_(0.5)
        pb[0].add_(1.0)
    assert AF._wgen["owner_gen"] == AF._cast_generation()
    cat2 = AF._w_bf16_cat([pb[2], pb[3], pb[4]], False)
    assert cat2 is cat and torch.equal(cat.cpu(), torch.cat([pb[2], pb[3], pb[4]]).detach().cpu().to(torch.bfloat16))
    assert torch.equal(AF._w_bf16(pb[0], False).cpu(), pb[0].detach().cpu().to(torch.bfloat16))
    # a weight registered later changes the cache generation: the claim lapses until the next optimizer step
    AF._w_bf16(pb[1], False)
    assert AF._wgen["owner_gen"] != AF._cast_generation()
    for q in pb:
        q.grad = torch.ones_like(q)
    ob.step()
    assert AF._wgen["owner_gen"] == AF._cast_generation()
    assert torch.equal(AF._w_bf16(pb[1], False).cpu(), pb[1].detach().cpu().to(torch.bfloat16))
    AF.invalidate_weight_cache()


def test_fused_adamw_writes_f16_forward_copies(dev):
    """Mixed mode: the f16 forward copies (functional._w_h16 / _w_h16_cat: dense weights and row slices of a fused Q/K/V buffer)
    of weights the tile pass updates are rewritten in that pass; the per-step refresh then only re-casts f16 copies the optimizer
    does not cover (a weight without any bf16 copy, e.g. linear_pos: updated by the linear kernel)."""
    from auto_avsr_amd import functional as AF

    torch.manual_seed(6)
    AF.invalidate_weight_cache()
    shapes = [(128, 64), (64, 64), (64, 64), (64, 64), (64, 128), (64,)]
    ps = [torch.nn.Parameter(torch.randn(s).to(dev)) for s in shapes]
    opt = FusedAdamW(ps, lr=1e-2, weight_decay=0.03, max_grad_norm=1.0, cast_weights=True)
    for w in (ps[0], ps[1], ps[2], ps[3]):
        AF._w_bf16(w, True)               # (the data-gradient copies every Linear of the model has)
    h0 = AF._w_h16(ps[0])
    hcat = AF._w_h16_cat((ps[1], ps[2], ps[3]))
    h4 = AF._w_h16(ps[4])                 # no bf16 copy registered: not in the tile plan
    for it in range(2):
        for q in ps:
            q.grad = torch.randn_like(q)
        opt.step()
        # two-plane images [out][2][in]: hi = f16(w), lo = f16((w - hi) * 2^11) (csrc/prims.h f2h_lo)
        def planes(w):
            w = w.detach().cpu()
            hi = w.half()
            return torch.stack([hi, ((w - hi.float()) * 2048.0).half()], 1)

        assert torch.equal(h0.cpu(), planes(ps[0]))
        assert torch.equal(hcat.cpu(), planes(torch.cat([ps[1], ps[2], ps[3]])))
        assert AF._wh16_owned == {(w.data_ptr(), tuple(w.shape)) for w in ps[:4]}
        assert not torch.equal(h4.cpu(), planes(ps[4]))  # stale until the refresh
        calls = []
        orig = AF.ops.multi_cast_transpose
        AF.ops.multi_cast_transpose = lambda t, n, b: (calls.append(n), orig(t, n, b))
        try:
            with AF.numerics("mixed"):
                AF.refresh_weight_cache()
        finally:
            AF.ops.multi_cast_transpose = orig
        assert calls == [1], calls        # ONE table entry: the uncovered weight
        assert torch.equal(h4.cpu(), planes(ps[4]))
        assert AF._w_h16(ps[0]) is h0 and AF._w_h16(ps[4]) is h4
    AF.invalidate_weight_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("cast", [False, True], ids=["plain", "cast_weights"])
def test_fused_adamw_under_hipgraph(cast):
    """The whole training step (forward, backward, fused optimizer, bf16 weight re-cast) captured once and replayed:
    parameters after three replays equal three eager steps (the step count, learning rate and clip coefficient live on
    the device, the gradient pointer table is re-sent from its pinned buffer by a captured copy node)."""
    from auto_avsr_amd import functional as AF

    dev = torch.device("cuda:0")

    def make():
        torch.manual_seed(3)
        l1, l2 = torch.nn.Linear(64, 128).to(dev), torch.nn.Linear(128, 64).to(dev)
        return l1, l2, FusedAdamW(list(l1.parameters()) + list(l2.parameters()), lr=1e-2, weight_decay=0.03, max_grad_norm=1.0,
                                 warmup_steps=2, total_steps=8, cast_weights=cast)

    x = torch.randn(32, 64, device=dev).bfloat16()

    def step(l1, l2, opt):
        AF.new_step()
        AF.refresh_weight_cache()
        h = AF.linear(x, l1.weight, l1.bias)
        y = AF.linear(h, l2.weight, l2.bias, out_dtype=torch.float32)
        (y * y).mean().backward()
        opt.step()

    AF.invalidate_weight_cache()
    work = torch.cuda.Stream()
    work.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(work):
        e1, e2, eo = make()
        for _ in range(3):
            step(e1, e2, eo)
            eo.zero_grad()
        AF.invalidate_weight_cache()
        g1, g2, go = make()
        step(g1, g2, go)  # warm-up (registers the weight copies), then rewind parameters and optimizer state
        go.zero_grad()
        torch.manual_seed(3)
        r1, r2 = torch.nn.Linear(64, 128).to(dev), torch.nn.Linear(128, 64).to(dev)
        with torch.no_grad():
            for p, q in zip(list(g1.parameters()) + list(g2.parameters()), list(r1.parameters()) + list(r2.parameters())):
                p.copy_(q)
            for t in go.exp_avg + go.exp_avg_sq:
                t.zero_()
            go.state.zero_()
        AF.claim_weight_casts(None, None)  # the parameters were rewound behind the optimizer's back
        AF.refresh_weight_cache()  # builds the cast table (H2D copy) outside the capture
    torch.cuda.current_stream().wait_stream(work)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(g1, g2, go)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert go.step_count == 3 and abs(go.last_lr - eo.last_lr) < 1e-9
    for p, q in zip(list(e1.parameters()) + list(e2.parameters()), list(g1.parameters()) + list(g2.parameters())):
        assert (p.detach() - q.detach()).abs().max() < 1e-5 * max(1.0, float(p.detach().abs().max()))
    AF.invalidate_weight_cache()
