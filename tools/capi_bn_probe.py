"""Probe: the cross-rank BatchNorm statistics path (bn_stats -> all-gather -> bn_finalize) through the C-API communicator,
eager vs captured into a hipGraph, one rank."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import functional as AF
from auto_avsr_amd import ops
from auto_avsr_amd.comm import StreamComm

dev = torch.device("cuda:0")
comm = StreamComm.single()
rows, C = 1000, 64
x = torch.randn(rows, C, device=dev).bfloat16()
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)


def stats(use_comm):
    AF.set_bn_sync(True if use_comm else None, comm=comm if use_comm else None)
    out = AF._bn_train_stats(x, rows, C, 1e-5, 0.1, rm, rv)
    AF.set_bn_sync(None)
    return out


def raw():
    mine = ops.bn_stats(x, rows, C, with_count=True)
    flat = torch.empty(mine.numel(), dtype=torch.float32, device=dev)
    comm.all_gather(flat, mine)
    return mine, flat


m0, i0, _ = stats(False)
m1, i1, n1 = stats(True)
torch.cuda.synchronize()
print("eager: mean diff", float((m0 - m1).abs().max()), "invstd diff", float((i0 - i1).abs().max()), "n", float(n1))
a, b = raw()
torch.cuda.synchronize()
print("eager raw gather equal:", bool(torch.equal(a, b)))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    raw()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ga, gb = raw()
    gm, gi, gn = stats(True)
g.replay()
torch.cuda.synchronize()
print("graph raw gather equal:", bool(torch.equal(ga, gb)), "nan in gathered:", bool(torch.isnan(gb).any()), "| mine[:4]", ga[:4].tolist(), "flat[:4]", gb[:4].tolist())
print("graph: mean diff", float((m0 - gm).abs().max()), "invstd diff", float((i0 - gi).abs().max()), "n", float(gn))
comm.close()
