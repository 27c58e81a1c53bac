"""Throughput of the audio-visual composition (auto_avsr_amd/e2e_av.py) at BASELINE config 5 (max-frames 3200):
full training step (fwd + bwd + fused clip / AdamW), hipGraph replay (--no-graph: eager), one batch shape.  Not part of bench.py's contract
(the reference snapshot has no AV model to compare with).  GPU box:  python tools/bench_av.py [--frames 3200] [--steps 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e_av import E2EAV
from auto_avsr_amd.optim import FusedAdamW


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3200)
    ap.add_argument("--T", type=int, default=400, help="frames per utterance (SURVEY 8d: max-frames 3200 => (8, 400, 64))")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--mode", choices=["bf16", "mixed", "hpf"], default="mixed", help="numerical mode (functional.set_mode), as bench.py")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    odim = 5049
    model = E2EAV(odim).to(dev).train()
    AF.set_mode(args.mode)
    AF.manual_seed(1)
    seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    AF.set_seed_tensor(seed_dev)
    opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0,
                     warmup_steps=5000, total_steps=75000, cast_weights=True)
    B, T = args.frames // args.T, args.T
    video = torch.randn(B, T, 1, 88, 88, device=dev)
    audio = torch.randn(B, T * 640, 1, device=dev)
    lengths = torch.full((B,), T, dtype=torch.int64, device=dev)
    label = torch.randint(1, odim - 1, (B, 1, max(1, round(T / 6.5))), device=dev)

    def step():
        AF.new_step()
        seed_dev.add_(1)
        AF.refresh_weight_cache()
        loss = model.forward_tensors(video, audio, lengths, label)[0]
        loss.backward()
        opt.step()
        opt.zero_grad()

    run = step
    if not args.no_graph:  # the same capture protocol as bench.py: one eager step on a side stream, then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        AF.refresh_weight_cache()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        run = g.replay
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    n = sum(p.numel() for p in model.parameters())
    print(f"E2EAV {n / 1e6:.1f} M parameters, batch {B} x {T} frames: {dt * 1e3:.2f} ms / step = {B * T / dt:,.0f} frames/s")
    import json

    print(json.dumps({"metric": "AV-fusion frames/sec (parallel audio + video encoders, shared decoder), full training step",
                      "value": round(B * T / dt, 1), "unit": "video-frames/sec", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2),
                      "dtype": {"bf16": "bf16", "mixed": "f16 / split-bf16 forward per component, bf16 backward", "hpf": "split-bf16 forward, bf16 backward"}[args.mode],
                      "data": "synthetic",
                      "config": {"workload": f"configs[4] single-GPU leg: E2EAV {n / 1e6:.0f} M parameters, max-frames {args.frames} "
                                             f"=> batch ({B}, {T}, {label.shape[2]}), " + ("eager launches" if args.no_graph else "hipGraph replay"),
                                 "parity": "n/a (no AV model in the reference snapshot, SURVEY F4)"}}))


if __name__ == "__main__":
    main()
