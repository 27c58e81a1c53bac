"""Golden vectors for the `espnet.nets.ctc_prefix_score` boundary classes: runs the REFERENCE CTCPrefixScoreTH with a
batch of two ragged utterances (windowing off), with and without pre-selected candidates, through two decoding steps
joined by `index_select_state`, and the host-side CTCPrefixScore for one hypothesis.
Run in the build container only:   python tests/golden/make_golden_prefix.py   ->  tests/golden/golden_prefix_v1.pt"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from espnet.nets.ctc_prefix_score import CTCPrefixScore, CTCPrefixScoreTH  # noqa: E402


def th_case(seed, B, T, V, W, S, xlens):
    g = torch.Generator().manual_seed(4000 + seed)
    logp = torch.log_softmax(torch.randn(B, T, V, generator=g) * 2.0, -1)
    eos = V - 1
    impl = CTCPrefixScoreTH(logp.clone(), torch.tensor(xlens), 0, eos)
    y0 = [[eos] for _ in range(B * W)]
    ids0 = None if S == 0 else torch.stack([torch.randperm(V - 2, generator=g)[:S] + 1 for _ in range(B * W)])
    sc0, st0 = impl(y0, None, ids0)
    # beam pruning: per utterance, W (hypothesis, token) pairs out of its W x V table
    masked = sc0.clone()
    masked[:, 0] = -1e30
    masked[:, eos] = -1e30
    if ids0 is not None:  # only scored candidates can be chosen
        keep = torch.zeros_like(masked, dtype=torch.bool)
        keep.scatter_(1, ids0, True)
        masked[~keep] = -1e30
    best = masked.view(B, W * V).topk(W, dim=1)[1]
    state1 = impl.index_select_state(st0, best)
    toks = torch.fmod(best, V).view(-1)
    y1 = [[eos, int(t)] for t in toks]
    ids1 = None if S == 0 else torch.stack([torch.randperm(V - 2, generator=g)[:S] + 1 for _ in range(B * W)])
    if ids1 is not None:
        for n in range(B * W):  # exercise the "candidate == last token" branch
            if not (ids1[n] == toks[n]).any():
                ids1[n, 0] = toks[n]
    sc1, st1 = impl(y1, state1, ids1)
    return dict(seed=seed, B=B, T=T, V=V, W=W, S=S, xlens=xlens, logp=logp, ids0=ids0, sc0=sc0, best=best,
                r_sel=state1[0].clone(), s_sel=state1[1].clone(), ids1=ids1, sc1=sc1, r1_sum=st1[0].double().clamp_min(-1e9).sum(),
                r1_sample=st1[0][:, :, ::2, ::3].clone())


def host_case(seed, T, V, S):
    g = torch.Generator().manual_seed(4100 + seed)
    logp = torch.log_softmax(torch.randn(T, V, generator=g) * 2.0, -1).numpy()
    impl = CTCPrefixScore(logp, 0, V - 1, np)
    r0 = impl.initial_state()
    cs0 = (torch.randperm(V - 1, generator=g)[:S] + 0).numpy()
    psi0, st0 = impl([V - 1], cs0, r0)
    pick = 1
    cs1 = np.concatenate([cs0[pick:pick + 1], (torch.randperm(V, generator=g)[: S - 1]).numpy()])
    psi1, st1 = impl([V - 1, int(cs0[pick])], cs1, st0[pick])
    return dict(seed=seed, T=T, V=V, S=S, logp=torch.from_numpy(logp), r0=torch.from_numpy(r0), cs0=torch.from_numpy(cs0),
                psi0=torch.from_numpy(psi0), pick=pick, cs1=torch.from_numpy(cs1), psi1=torch.from_numpy(psi1),
                st1=torch.from_numpy(np.ascontiguousarray(st1)))


if __name__ == "__main__":
    out = {"th": [th_case(1, 2, 14, 21, 3, 6, [14, 10]), th_case(2, 2, 11, 17, 2, 0, [9, 11]), th_case(3, 1, 19, 30, 4, 8, [19])],
           "host": [host_case(1, 13, 18, 5), host_case(2, 22, 26, 7)]}
    torch.save(out, os.path.join(HERE, "golden_prefix_v1.pt"))
    for c in out["th"]:
        print(c["seed"], c["sc0"].shape, float(c["sc1"].clamp_min(-1e9).sum()))
