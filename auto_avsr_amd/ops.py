"""Thin Python wrappers over the C ABI (include/avsr_hip.h): one function per entry point,
taking torch tensors as device-memory handles.  No compute happens in Python or ATen here."""
import torch

from . import _lib

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}
NT, NN, TN = 0, 1, 2


def dt(t):
    return _DT[t.dtype]


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(*ts):
    L = _lib.lib()
    for t in ts:
        if t is None:
            continue
        if t.is_cuda:
            if L.is_emulator:
                raise _lib.AvsrLibraryError("emulator build cannot take device tensors")
            return torch.cuda.current_stream(t.device).cuda_stream
        if not L.is_emulator:
            raise _lib.AvsrLibraryError(
                "libavsr_hip.so operates on GPU memory only: got a CPU tensor (no CPU fallback exists)"
            )
        return None
    return None


def call(name, *args):
    _lib.lib().call(name, *args)


def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-12):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("avsr_layernorm_fwd", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), dt(y), _ptr(mean), _ptr(rstd),
         rows, cols, eps, _stream(x))
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    call("avsr_layernorm_bwd", _ptr(dy), dt(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres),
         _ptr(dx), _ptr(dgamma), _ptr(dbeta), rows, cols, _stream(x))
    return dx


def gemm(layout, A, lda, B, ldb, M, N, K, C, ldc, *, precise=False, bias=None, act=0, gate=None, ldg=0,
         gate_scale=1.0, drop_p=0.0, seed=0, alpha=1.0, resid=None, ldr=0, accumulate=False, split_k=1,
         force_tile=0):
    call("avsr_gemm", layout, _ptr(A), dt(A), lda, _ptr(B), dt(B), ldb, M, N, K, int(precise), _ptr(bias), act,
         _ptr(gate), dt(gate) if gate is not None else 0, ldg, gate_scale, drop_p, seed, alpha, _ptr(resid), ldr,
         _ptr(C), dt(C), ldc, int(accumulate), split_k, force_tile, _stream(A))
    return C
