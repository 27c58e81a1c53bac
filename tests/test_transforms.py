"""Device-side input pipeline (csrc/augment.hip, auto_avsr_amd/transforms.py) against
(a) golden vectors produced by the REFERENCE's own datamodule/transforms.py / data_module.py
    (tests/golden/make_golden_transforms.py; torchvision / torchaudio calls stubbed with their restated algorithms), and
(b) the CPU restatement oracle/transforms_oracle.py on further seeded inputs.
Video results are required to be bit-identical (f32); audio within 2e-5 (different summation order of the statistics)."""
import os
import random
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import transforms_oracle as TO  # noqa: E402

from auto_avsr_amd import transforms as TR  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return torch.load(os.path.join(HERE, "golden", "golden_transforms_v1.pt"), weights_only=False)


def make_noise():
    return torch.randn(1, 96000, generator=torch.Generator().manual_seed(2024)) * 0.05


def make_clip(frames, seed):
    return torch.randint(0, 256, (frames, 96, 96, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed))


def make_wav(n, seed):
    return torch.randn(n, 1, generator=torch.Generator().manual_seed(seed)) * 0.1


def seed_all(seed):
    torch.manual_seed(seed)
    random.seed(seed)


# ---------------------------------------------------------------------------------------------- oracle vs reference
def test_oracle_matches_reference_golden(golden):
    assert torch.equal(make_noise()[0, ::9600], golden["noise_probe"])
    for tag in ("video_train", "video_train2", "video_val"):
        c = golden[tag]
        seed_all(c["seed"])
        out, _, _ = TO.video_transform(make_clip(c["frames"], c["seed"]).permute(0, 3, 1, 2), c["subset"])
        assert tuple(out.shape) == c["shape"] and torch.equal(out[::12], c["sample_frames"])
        # whole-tensor checksums (f64 sums: the reduction order depends on the host's thread count, hence the tolerance)
        assert abs(float(out.double().sum()) - c["sum"]) < 1e-7 and abs(float(out.double().abs().sum()) - c["abssum"]) < 1e-7
    for m in golden["masks"]:
        seed_all(m["seed"])
        ivs = TO.adaptive_time_mask_intervals(m["length"], m["window"], m["stride"])
        z = torch.zeros(m["length"], dtype=torch.bool)
        for a, b in ivs:
            z[a:b] = True
        assert torch.equal(z.nonzero().flatten(), m["zero"])
    noise = make_noise()
    for tag in ("audio_train", "audio_train2", "audio_val", "audio_val_snr"):
        c = golden[tag]
        seed_all(c["seed"])
        use_noise = c["subset"] == "train" or c["snr_target"] is not None
        out, _, _, _ = TO.audio_transform(make_wav(c["n"], c["seed"]), c["subset"], noise if use_noise else None, c["snr_target"])
        assert torch.equal(out[::4], c["out_every4"])
    p = golden["pad"]
    b, lens = TO.pad([torch.arange(5.0).view(5, 1), torch.arange(3.0).view(3, 1), torch.arange(4.0).view(4, 1)], 0.0)
    assert torch.equal(b, p["batch"]) and lens == p["lengths"]
    tb, tl = TO.pad([torch.tensor([3, 4, 5]), torch.tensor([7])], -1)
    assert torch.equal(tb, p["target_batch"]) and tl == p["target_lengths"]


# ---------------------------------------------------------------------------------------------- product vs reference
def test_video_transform_matches_reference_golden(dev, golden):
    """Bit-identical f32 output, single clips through the reference-named class."""
    for tag in ("video_train", "video_train2", "video_val"):
        c = golden[tag]
        seed_all(c["seed"])
        clip = make_clip(c["frames"], c["seed"]).to(dev)
        out = TR.VideoTransform(c["subset"])(clip.permute(0, 3, 1, 2)).cpu()  # load_video's [T, 3, H, W] view
        assert tuple(out.shape) == c["shape"] and torch.equal(out[::12], c["sample_frames"])
        # whole-tensor checksums (f64 sums: the reduction order depends on the host's thread count, hence the tolerance)
        assert abs(float(out.double().sum()) - c["sum"]) < 1e-7 and abs(float(out.double().abs().sum()) - c["abssum"]) < 1e-7


def test_adaptive_time_mask_protocol(golden):
    for m in golden["masks"]:
        seed_all(m["seed"])
        ivs = TR.AdaptiveTimeMask(m["window"], m["stride"]).draw(m["length"])
        z = torch.zeros(m["length"], dtype=torch.bool)
        for a, b in ivs:
            z[a:b] = True
        assert torch.equal(z.nonzero().flatten(), m["zero"])


def test_audio_transform_matches_reference_golden(dev, golden):
    noise = make_noise()
    for tag in ("audio_train", "audio_train2", "audio_val", "audio_val_snr"):
        c = golden[tag]
        seed_all(c["seed"])
        use_noise = c["subset"] == "train" or c["snr_target"] is not None
        tr = TR.AudioTransform(c["subset"], snr_target=c["snr_target"], noise=noise.to(dev) if use_noise else None)
        out = tr(make_wav(c["n"], c["seed"]).to(dev)).cpu()
        assert out.shape == (c["n"], 1)
        assert (out[::4] - c["out_every4"]).abs().max() < 2e-5
        assert abs(float(out.double().sum()) - c["sum"]) < 1e-2 and abs(float((out.double() ** 2).sum()) - c["sqsum"]) < 1e-2 * c["n"]


# ---------------------------------------------------------------------------------------------- batches vs oracle
@pytest.mark.parametrize("subset", ["train", "val"])
def test_video_batch_collation_vs_oracle(dev, subset):
    """Ragged batch in one launch: same RNG consumption as transforming the clips one after the other, zero padding as
    collate_pad, bit-identical values; bf16 output = the rounded f32 output."""
    lens = [37, 9, 52, 26]
    clips = [make_clip(n, 100 + i) for i, n in enumerate(lens)]
    seed_all(77)
    ref = [TO.video_transform(c.permute(0, 3, 1, 2), subset)[0] for c in clips]
    want, want_lens = TO.pad(ref, 0.0)
    seed_all(77)
    got, got_lens = TR.video_batch([c.to(dev) for c in clips], subset)
    assert got_lens == want_lens and torch.equal(got.cpu(), want)
    seed_all(77)
    got16, _ = TR.video_batch([c.to(dev) for c in clips], subset, out_dtype=torch.bfloat16)
    assert torch.equal(got16.cpu(), want.bfloat16())
    if subset == "train":  # the masked frames are there and carry (0 - mean) / std
        assert (want == (0.0 - 0.421) / 0.165).all(dim=(2, 3, 4)).any()


def test_audio_batch_collation_vs_oracle(dev):
    noise = make_noise()
    lens = [30000, 16000, 47000]
    wavs = [make_wav(n, 200 + i) for i, n in enumerate(lens)]
    for subset, snr_target in (("train", None), ("val", None), ("val", 10)):
        use_noise = subset == "train" or snr_target is not None
        seed_all(5)
        ref = [TO.audio_transform(w, subset, noise if use_noise else None, snr_target)[0] for w in wavs]
        want, want_lens = TO.pad(ref, 0.0)
        seed_all(5)
        an = TR.AddNoise(noise=noise.to(dev), snr_target=None if subset == "train" else snr_target) if use_noise else None
        got, got_lens = TR.audio_batch([w.to(dev) for w in wavs], subset, an)
        assert got_lens == want_lens and got.shape == want.shape
        assert (got.cpu() - want).abs().max() < 2e-5
        for i, n in enumerate(lens):
            assert not got[i, n:].any()


def test_pad_targets():
    out, lens = TR.pad_targets([torch.tensor([3, 4, 5]), torch.tensor([7])])
    want, wl = TO.pad([torch.tensor([3, 4, 5]), torch.tensor([7])], -1)
    assert torch.equal(out, want) and lens == wl


# ---------------------------------------------------------------------------------------------- third-party known answers
def _third_party():
    import json

    return json.load(open(os.path.join(HERE, "golden", "golden_thirdparty_v1.json")))


def test_third_party_ops_oracle_known_answers():
    """The five torchvision / torchaudio operations of the reference's input path (absent from the reference tree and from this
    image) against known-answer vectors computed from their PUBLISHED definitions with exact rational arithmetic
    (tests/golden/make_golden_thirdparty.py) -- the oracle's restatements, one by one."""
    g = _third_party()
    rgb = torch.tensor(g["rgb"], dtype=torch.float32).view(-1, 3, 1, 1) / 255.0
    gray = TO.rgb_to_grayscale(rgb).flatten()
    assert (gray - torch.tensor(g["gray"])).abs().max() < 2e-7
    assert (TO.normalize(gray, 0.421, 0.165) - torch.tensor(g["gray_normalized"])).abs().max() < 2e-6
    for c in g["center_crop"]:
        assert TO.center_crop_params(c["H"], c["W"], c["size"]) == (c["top"], c["left"])
    for c in g["random_crop"]:
        torch.manual_seed(c["seed"])
        assert TO.random_crop_params(c["H"], c["W"], c["size"]) == (c["i"], c["j"])
    for c in g["add_noise"]:
        y = TO.add_noise(torch.tensor([c["x"]], dtype=torch.float32), torch.tensor([c["n"]], dtype=torch.float32),
                         torch.tensor([float(c["snr_db"])]))
        assert (y[0] - torch.tensor(c["y"])).abs().max() < 1e-5 * max(abs(v) for v in c["y"])


def test_third_party_ops_kernels_known_answers(dev):
    """The same known answers through the PRODUCT (csrc/augment.hip): constant-colour clips through the evaluation pipeline
    (CenterCrop -> Grayscale -> Normalize, transforms.py:100-105) must give the hand-computed luma, normalised; the train
    pipeline's crop origin must be RandomCrop.get_params's draw (a clip whose pixel values encode their own coordinates);
    AddNoise at a fixed SNR on a constant noise recording must give add_noise's closed form (before the final layer norm,
    which is undone here with the known mean / variance of the expected signal)."""
    g = _third_party()
    for (r, gg, b), want in zip(g["rgb"], g["gray_normalized"]):
        clip = torch.tensor([r, gg, b], dtype=torch.uint8).view(1, 1, 1, 3).expand(2, 96, 96, 3).contiguous()
        out = TR.VideoTransform("val")(clip.to(dev).permute(0, 3, 1, 2)).cpu()
        assert tuple(out.shape) == (2, 1, 88, 88) and (out - want).abs().max() < 2e-6, (r, gg, b)
    # crop origin: red channel = row index, green = column index, blue = 0  ->  luma(i + y, j + x) identifies (i, j)
    yy, xx = torch.meshgrid(torch.arange(96), torch.arange(96), indexing="ij")
    clip = torch.stack([yy, xx, torch.zeros_like(yy)], dim=-1).to(torch.uint8).unsqueeze(0).contiguous()
    for c in g["random_crop"]:
        seed_all(c["seed"])
        torch.manual_seed(c["seed"])
        out = TR.VideoTransform("train")(clip.to(dev).permute(0, 3, 1, 2)).cpu()[0, 0]
        luma = out * 0.165 + 0.421
        if float(out.abs().max()) and not bool((out == (0.0 - 0.421) / 0.165).all()):  # (a one-frame clip may be time-masked)
            want00 = (0.2989 * c["i"] + 0.587 * c["j"]) / 255.0
            assert abs(float(luma[0, 0]) - want00) < 1e-5, (c, float(luma[0, 0]), want00)
    for c in g["add_noise"]:
        x = torch.tensor(c["x"], dtype=torch.float32).view(-1, 1)
        n = torch.tensor([c["n"]], dtype=torch.float32)
        seed_all(0)
        tr = TR.AudioTransform("val", snr_target=c["snr_db"] if c["snr_db"] else None, noise=n.to(dev))
        if not c["snr_db"]:  # snr_target = 0 means "none" to the reference's truthiness test (transforms.py:72): plain layer norm
            continue
        out = tr(x.to(dev)).cpu().flatten()
        y = torch.tensor(c["y"], dtype=torch.float64)
        want = (y - y.mean()) / torch.sqrt(y.var(unbiased=False) + 1e-8)
        assert (out.double() - want).abs().max() < 2e-5
