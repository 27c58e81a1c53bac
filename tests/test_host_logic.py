"""Host-side logic that surrounds the hot path: masks / padding helpers (with the reference's docstring
known-answers), the LR schedule, checkpoint averaging, the synthetic bucketed workload, and the C-ABI surface."""
import ctypes
import math
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from auto_avsr_amd import _lib, nets  # noqa: E402
from auto_avsr_amd.synthetic import bucket_batches, make_batch, rank_batches, utterance_lengths  # noqa: E402


def test_reference_docstring_known_answers():
    # nets_utils.py:44-51 (pad_list), :79-147 (make_pad_mask), :198-266 (make_non_pad_mask), mask.py:18-21
    x = [torch.ones(4), torch.ones(2), torch.ones(1)]
    assert nets.pad_list(x, 0).tolist() == [[1, 1, 1, 1], [1, 1, 0, 0], [1, 0, 0, 0]]
    assert nets.make_pad_mask([5, 3, 2]).int().tolist() == [[0, 0, 0, 0, 0], [0, 0, 0, 1, 1], [0, 0, 1, 1, 1]]
    assert nets.make_non_pad_mask([5, 3, 2]).int().tolist() == [[1, 1, 1, 1, 1], [1, 1, 1, 0, 0], [1, 1, 0, 0, 0]]
    xs = torch.zeros((3, 2, 4))
    assert nets.make_pad_mask([5, 3, 2], xs)[1].int().tolist() == [[0, 0, 0, 1], [0, 0, 0, 1]]
    assert nets.subsequent_mask(3).int().tolist() == [[1, 0, 0], [1, 1, 0], [1, 1, 1]]


def test_add_sos_eos_static_equals_reference_semantics():
    y = torch.tensor([[[5, 6, 7, -1, -1]], [[1, 2, 3, 4, 9]], [[8, -1, -1, -1, -1]]])
    sos = eos = 99
    ys_in, ys_out = nets.add_sos_eos(y, sos, eos, -1)           # add_sos_eos.py:12-31 semantics
    s_in, s_out = nets.add_sos_eos_static(y, sos, eos, -1)      # static width, no host sync
    assert ys_in.tolist() == [[99, 5, 6, 7, 99, 99], [99, 1, 2, 3, 4, 9], [99, 8, 99, 99, 99, 99]]
    assert ys_out.tolist() == [[5, 6, 7, 99, -1, -1], [1, 2, 3, 4, 9, 99], [8, 99, -1, -1, -1, -1]]
    assert s_in.tolist() == ys_in.tolist() and s_out.tolist() == ys_out.tolist()
    # a batch where every row is padded: the static version keeps the extra (ignored) column
    y2 = torch.tensor([[[5, -1, -1]], [[1, 2, -1]]])
    a_in, a_out = nets.add_sos_eos(y2, sos, eos, -1)
    b_in, b_out = nets.add_sos_eos_static(y2, sos, eos, -1)
    assert b_in[:, : a_in.shape[1]].tolist() == a_in.tolist() and b_out[:, : a_out.shape[1]].tolist() == a_out.tolist()
    assert (b_out[:, a_out.shape[1]:] == -1).all()


def test_warmup_cosine_schedule():
    sys.path.insert(0, ROOT)
    from cosine import WarmupCosineScheduler

    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([w], lr=1.0)
    sched = WarmupCosineScheduler(opt, warmup_epochs=2, total_epochs=10, steps_per_epoch=5)
    lrs = []
    for _ in range(50):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    # cosine.py:20-25: linear warm-up over 10 steps, then 0.5*(1+cos(pi*(s-10)/40))
    for s, lr in enumerate(lrs, start=1):
        ref = s / 10 if s < 10 else 0.5 * (1 + math.cos(math.pi * (s - 10) / 40))
        assert abs(lr - ref) < 1e-6, (s, lr, ref)


def test_average_checkpoints(tmp_path):
    from average_checkpoints import average_checkpoints

    paths = []
    for i in range(3):
        sd = {"model.w": torch.full((2,), float(i)), "model.n": torch.tensor(3 * i), "other": torch.ones(1)}
        p = tmp_path / f"epoch={i}.ckpt"
        torch.save({"state_dict": sd}, p)
        paths.append(str(p))
    avg = average_checkpoints(paths)
    assert set(avg) == {"w", "n"} and avg["w"].tolist() == [1.0, 1.0] and int(avg["n"]) == 3


def _reference_batching(lengths, max_frames, num_buckets):
    """Restatement of datamodule/data_module.py:44-62,79-99 used as the expected value."""
    lt = torch.tensor(lengths)
    edges = torch.linspace(float(lt.min()), float(lt.max()), num_buckets)
    assign = torch.bucketize(lt, edges)
    items = sorted([(i, int(l), int(assign[i])) for i, l in enumerate(lengths)], key=lambda t: t[1], reverse=True)
    items = sorted(items, key=lambda t: t[2])
    batches, cur, tot = [], [], 0
    for i, l, _ in items:
        if tot + l > max_frames:
            batches.append(cur)
            cur, tot = [i], l
        else:
            cur.append(i)
            tot += l
    if cur:
        batches.append(cur)
    return batches


def test_bucketed_batches_follow_reference_algorithm():
    lengths = utterance_lengths(n=2000, seed=7)
    assert lengths.min() >= 12 and lengths.max() <= 400
    got = bucket_batches(lengths, 1600, 400)
    assert got == _reference_batching(lengths.tolist(), 1600, 400)
    assert all(sum(int(lengths[i]) for i in b) <= 1600 for b in got)
    assert sorted(i for b in got for i in b) == list(range(2000))
    r0, r1 = rank_batches(got, 0, 2, seed=3), rank_batches(got, 1, 2, seed=3)
    # DistributedSampler semantics: every rank gets ceil(n / world) batches (the tail is padded with the head of the
    # shuffled list), together they cover every batch
    assert len(r0) == len(r1) == (len(got) + 1) // 2
    assert {tuple(b) for b in r0} | {tuple(b) for b in r1} == {tuple(b) for b in got}
    assert len({tuple(b) for b in r0} & {tuple(b) for b in r1}) <= len(got) % 2
    x, lens, y, frames = make_batch(lengths, got[len(got) // 2], "video", 5049, seed=1)
    assert x.shape[1] == int(lens.max()) and frames == int(lens.sum()) and y.shape[1] == 1
    assert (x[0, int(lens[0]):] == 0).all() and y.max() < 5048 and (y[y != -1] >= 1).all()


def test_c_abi_library_exports_every_declared_symbol():
    """include/avsr_hip.h is the contract: the gfx950 library must export each prototype (no compute calls here)."""
    from auto_avsr_amd import build

    lib = build.build_hip()
    protos = _lib.parse_header()
    assert len(protos) >= 40
    exported = set(subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout.split())
    assert not [n for n in protos if n not in exported]
    # and the binding refuses to run device kernels on host memory (no CPU fallback in the product)
    _lib._lib = None
    L = _lib.lib()
    assert not L.is_emulator
    from auto_avsr_amd import ops

    with pytest.raises(_lib.AvsrLibraryError):
        ops.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), torch.float32)
    _lib._lib = None


def test_state_dict_contract():
    from auto_avsr_amd.e2e import E2E

    m = E2E(5049, "video")
    sd = m.state_dict()
    assert len(sd) == 767 and sum(p.numel() for p in m.parameters()) == 250_383_410
    for k, shape in {
        "frontend.frontend3D.0.weight": (64, 1, 5, 7, 7),
        "encoder.encoders.0.conv_module.pointwise_cov1.weight": (1536, 768, 1),
        "encoder.encoders.11.conv_module.depthwise_conv.weight": (768, 1, 31),
        "encoder.encoders.3.self_attn.pos_bias_u": (12, 64),
        "decoder.embed.0.weight": (5049, 768),
        "decoder.decoders.5.src_attn.linear_out.bias": (768,),
        "ctc.ctc_lo.weight": (5049, 768),
    }.items():
        assert tuple(sd[k].shape) == shape, k
    assert sum(p.numel() for p in E2E(5049, "audio").parameters()) == 243_049_202


def test_qkv_bias_packing():
    """The three projection biases of an attention module are thirds of one buffer (no per-step concatenation for the fused
    Q/K/V projection); names / shapes / state_dict are unaffected; a layout lost by deepcopy falls back to torch.cat."""
    import copy

    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.nets import MultiHeadedAttention

    m = MultiHeadedAttention(2, 128, 0.0)
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(torch.float32)  # any _apply: parameters get fresh storages, then are packed again
    bq, bk, bv = m.linear_q.bias, m.linear_k.bias, m.linear_v.bias
    assert bk.data_ptr() == bq.data_ptr() + 4 * 128 and bv.data_ptr() == bq.data_ptr() + 8 * 128
    cat = AF._bias3(bq, bk, bv)
    assert cat.data_ptr() == bq.data_ptr() and torch.equal(cat, torch.cat([sd["linear_q.bias"], sd["linear_k.bias"], sd["linear_v.bias"]]))
    assert set(m.state_dict()) == set(sd) and all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    with torch.no_grad():
        bk.add_(1.0)  # an in-place update of one parameter is seen through the packed view
    assert torch.equal(AF._bias3(bq, bk, bv)[128:256], sd["linear_k.bias"] + 1.0)
    m2 = copy.deepcopy(m)
    c2 = AF._bias3(m2.linear_q.bias, m2.linear_k.bias, m2.linear_v.bias)
    assert torch.equal(c2, AF._bias3(bq, bk, bv))


def test_qkv_bias_packing_with_adjacent_separate_storages():
    """Three biases that merely SIT back to back in memory (separate storages: what a device allocator hands out for three
    small consecutive allocations) must still be packed into one storage -- adjacency alone once made the packer return
    early and every fused projection pay a torch.cat."""
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.nets import MultiHeadedAttention

    m = MultiHeadedAttention(2, 128, 0.0)
    arena = torch.randn(3 * 128)
    vals = [arena[i * 128:(i + 1) * 128].clone() for i in range(3)]
    # emulate "adjacent but separate": each bias a view of the arena is ONE storage, so fake the storage test instead
    for lin, i in ((m.linear_q, 0), (m.linear_k, 1), (m.linear_v, 2)):
        lin.bias.data = vals[i]
    real = torch.Tensor.data_ptr
    base = 1 << 20
    fake = {id(m.linear_q.bias): base, id(m.linear_k.bias): base + 512, id(m.linear_v.bias): base + 1024}
    try:
        torch.Tensor.data_ptr = lambda t: fake.get(id(t), real(t))
        m._pack_qkv_bias()
    finally:
        torch.Tensor.data_ptr = real
    bq, bk, bv = m.linear_q.bias, m.linear_k.bias, m.linear_v.bias
    assert bq.untyped_storage().data_ptr() == bv.untyped_storage().data_ptr()
    assert AF._bias3(bq, bk, bv).data_ptr() == bq.data_ptr()
    assert torch.equal(AF._bias3(bq, bk, bv), torch.cat(vals))


def test_model_module_fit_hooks_set_the_benchmarked_numerics():
    """lightning.ModelModule under a Trainer: on_fit_start puts the hot path into train.py's --numerics mode (default: mixed, the
    benchmarked one), on_fit_end restores what was there (the entry points ran in a mode nothing had benchmarked)."""
    import types

    import lightning as LM
    from auto_avsr_amd import functional as AF

    mod = LM.ModelModule.__new__(LM.ModelModule)
    for numerics, want in ((None, "mixed"), ("hpf", "hpf")):
        mod.args = types.SimpleNamespace(numerics=numerics)
        assert AF.mode() == "bf16"
        LM.ModelModule.on_fit_start(mod)
        assert AF.mode() == want
        LM.ModelModule.on_fit_end(mod)
        assert AF.mode() == "bf16" and AF._state["bn_sync"] is None
