"""Device-side input pipeline (csrc/augment.hip) at the bench batch geometry: 1600 frames per batch.
Prints achieved HBM GB/s (algorithmic bytes: 3 B read per cropped pixel + 4 B / 2 B written) and the CPU time of the
oracle restatement of the reference transforms on the same clips.  GPU box:  python tools/microbench_augment.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch

from auto_avsr_amd import transforms as TR

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for B, T in ((16, 100), (4, 400)):
    clips_cpu = [torch.randint(0, 256, (T, 96, 96, 3), dtype=torch.uint8) for _ in range(B)]
    clips = [c.to(dev) for c in clips_cpu]
    for dt, ob in ((torch.float32, 4), (torch.bfloat16, 2)):
        us = timeit(lambda: TR.video_batch(clips, "train", out_dtype=dt))
        byts = B * T * 88 * 88 * (3 + ob)
        print(f"video_batch B={B} T={T} out={dt}: {us:8.1f} us per batch (host table + launch), {byts / us * 1e-3:8.1f} GB/s algorithmic")
    # kernel only: replay the same launch without the host-side draws
    from auto_avsr_amd import ops
    rec = []
    ops.RECORD = ("avsr_video_transform", rec)
    out, _ = TR.video_batch(clips, "train")
    ops.RECORD = None
    args = rec[0][0]
    us = timeit(lambda: ops._lib.lib().call("avsr_video_transform", *args), iters=100)
    print(f"   kernel alone (f32 out): {us:8.1f} us, {B * T * 88 * 88 * 7 / us * 1e-3:8.1f} GB/s of ~8000 peak")
    wavs = [(torch.randn(T * 640, 1) * 0.1).to(dev) for _ in range(B)]
    an = TR.AddNoise(noise=(torch.randn(1, 16000 * 60) * 0.05).to(dev))
    us = timeit(lambda: TR.audio_batch(wavs, "train", an))
    print(f"audio_batch B={B} T={T * 640} samples: {us:8.1f} us per batch")
    if "--cpu" in sys.argv:
        import transforms_oracle as TO
        torch.set_num_threads(os.cpu_count() or 8)
        t0 = time.perf_counter()
        for c in clips_cpu:
            TO.video_transform(c.permute(0, 3, 1, 2), "train")
        t1 = time.perf_counter()
        print(f"   CPU oracle (reference transform chain, {torch.get_num_threads()} threads): {1e3 * (t1 - t0):8.1f} ms per batch = "
              f"{B * T / (t1 - t0):9.0f} frames/s")
