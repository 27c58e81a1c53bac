"""Synthetic LRS3-shaped workload (no data on the box): utterance lengths, length bucketing and greedy packing
by --max-frames restated from datamodule/data_module.py:44-106 (CustomBucketDataset / _batch_by_token_count),
padding from :10-41 (collate_pad).  Used by bench.py and the tests; SURVEY.md section 8(d)."""
import numpy as np
import torch


def utterance_lengths(n=20000, seed=42, lo=12, hi=400):
    """T_i ~ clip(round(LogNormal(ln 110, 0.6)), 12, 400) frames (25 fps; segments <= 16 s)."""
    rng = np.random.default_rng(seed)
    t = np.rint(rng.lognormal(np.log(110.0), 0.6, size=n))
    return np.clip(t, lo, hi).astype(np.int64)


def bucket_batches(lengths, max_frames=1600, num_buckets=400):
    """Batches of utterance indices: bucketize by length, order by (bucket, length desc), pack greedily while the
    sum of lengths stays <= max_frames (data_module.py:44-62,79-99)."""
    lengths_t = torch.as_tensor(lengths)
    assert max_frames >= int(lengths_t.max())
    edges = torch.linspace(float(lengths_t.min()), float(lengths_t.max()), num_buckets)
    bucket = torch.bucketize(lengths_t, edges).tolist()
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))  # length desc (stable)
    order = sorted(order, key=lambda i: bucket[i])  # then by bucket (stable)
    batches, cur, count = [], [], 0
    for i in order:
        n = int(lengths[i])
        if count + n > max_frames:
            batches.append(cur)
            cur, count = [i], n
        else:
            cur.append(i)
            count += n
    if cur:
        batches.append(cur)
    return batches


def make_batch(lengths, idxs, modality="video", odim=5049, seed=0, device="cpu", on_device=False):
    """One collated batch: inputs (B,T,1,88,88) [audio: (B,640T,1)] zero-padded, input_lengths (B,),
    targets (B,1,L) padded with -1; L_i = max(1, round(T_i/6.5)) ids uniform in [1, odim-2].
    on_device: draw the input samples with the DEVICE's generator (a training loop that makes a fresh batch per step cannot wait
    80 ms for 12 M host-side normal deviates + a 50 MB upload in front of a 22 ms step); labels / lengths as always."""
    g = torch.Generator().manual_seed(10007 * seed + 17)
    if on_device and torch.device(device).type != "cpu":
        return _make_batch_on_device(lengths, idxs, modality, odim, seed, torch.device(device), g)
    ts = [int(lengths[i]) for i in idxs]
    B, T = len(ts), max(ts)
    ls = [max(1, int(round(t / 6.5))) for t in ts]
    L = max(ls)
    y = torch.full((B, 1, L), -1, dtype=torch.int64)
    for b, n in enumerate(ls):
        y[b, 0, :n] = torch.randint(1, odim - 1, (n,), generator=g)
    if modality == "video":
        x = torch.zeros(B, T, 1, 88, 88)
        for b, t in enumerate(ts):
            x[b, :t] = torch.randn(t, 1, 88, 88, generator=g)
        lens = torch.tensor(ts, dtype=torch.int64)
    else:
        x = torch.zeros(B, T * 640, 1)
        for b, t in enumerate(ts):
            w = torch.randn(t * 640, generator=g)
            x[b, : t * 640, 0] = (w - w.mean()) / w.std()
        lens = torch.tensor(ts, dtype=torch.int64) * 640
    return x.to(device), lens.to(device), y.to(device), sum(ts)


def _make_batch_on_device(lengths, idxs, modality, odim, seed, dev, g):
    gd = torch.Generator(device=dev).manual_seed(10007 * seed + 17)
    ts = [int(lengths[i]) for i in idxs]
    B, T = len(ts), max(ts)
    ls = [max(1, int(round(t / 6.5))) for t in ts]
    y = torch.full((B, 1, max(ls)), -1, dtype=torch.int64)
    for b, n in enumerate(ls):
        y[b, 0, :n] = torch.randint(1, odim - 1, (n,), generator=g)
    tl = torch.tensor(ts, dtype=torch.int64)
    if modality == "video":
        x = torch.randn(B, T, 1, 88, 88, device=dev, generator=gd)
        x *= (torch.arange(T, device=dev)[None, :] < tl.to(dev)[:, None]).view(B, T, 1, 1, 1)  # zero padding
        lens = tl
    else:
        x = torch.randn(B, T * 640, device=dev, generator=gd)
        valid = torch.arange(T * 640, device=dev)[None, :] < (tl.to(dev) * 640)[:, None]
        x = x * valid
        n = (tl.to(dev) * 640).float()[:, None]
        mean = x.sum(1, keepdim=True) / n
        var = (((x - mean) * valid) ** 2).sum(1, keepdim=True) / (n - 1)
        x = (((x - mean) / var.sqrt()) * valid).unsqueeze(-1)
        lens = tl * 640
    return x, lens.to(dev), y.to(dev), sum(ts)


def rank_batches(batches, rank, world, seed=0):
    """DistributedSampler-style assignment: seeded shuffle of the batch list, padded with its own head to a multiple of
    `world` (so that EVERY rank gets the same number of batches -- a rank that ran out early would leave the others
    waiting in the gradient all-reduce), round-robin over ranks."""
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(len(batches), generator=g).tolist()
    pad = (-len(perm)) % world
    perm = perm + perm[:pad]
    return [batches[i] for i in perm[rank::world]]
