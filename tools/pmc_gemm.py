"""Workload for rocprofv3 --pmc passes: the three M = B*T GEMM shapes of a Conformer layer on the tuned NT kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(1600, 768, 768), (1600, 3072, 768), (1600, 768, 3072)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_bf16_nt(A, K, B, K, M, N, K, C, N)
    torch.cuda.synchronize()
