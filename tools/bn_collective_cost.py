"""GPU, one rank: what ONE cross-rank BatchNorm collective costs on the C-API communicator when nobody has to be waited for -- the
floor of the 61 + 64 latency-bound collectives per data-parallel step (a world-size-1 RCCL group: launch + kernel, no link time).
Back-to-back on one stream, eager and replayed from a hipGraph; beside it the same payloads as plain device copies.
-> gpurun_out/bn_collective_cost.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd.comm import StreamComm

dev = torch.device("cuda:0")
comm = StreamComm.single()
N = 200
res = {}
for C in (64, 256, 768):
    mine = torch.randn(3 * C + 1, device=dev)
    flat = torch.empty_like(mine)
    sums = torch.randn(2 * C, device=dev)

    def gather():
        for _ in range(N):
            comm.all_gather(flat, mine)

    def reduce():
        for _ in range(N):
            comm.all_reduce(sums)

    def copies():
        for _ in range(N):
            flat.copy_(mine)

    for name, fn in (("all_gather", gather), ("all_reduce", reduce), ("device_copy", copies)):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        eager = s.elapsed_time(e) / N * 1e3
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        res[f"C{C}_{name}"] = {"eager_us": round(eager, 2), "graph_us": round(s.elapsed_time(e) / N * 1e3, 2)}
        print(f"C={C} {name}: eager {eager:.2f} us, graph replay {res[f'C{C}_{name}']['graph_us']:.2f} us per call", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bn_collective_cost.json", "w"), indent=1)
