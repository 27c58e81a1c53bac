"""Golden vectors for the attention-branch target preparation: runs the REFERENCE's own add_sos_eos and target_mask
(espnet/nets/pytorch_backend/transformer/add_sos_eos.py:12-31, mask.py:27-37, called at e2e_asr_conformer.py:138-139) on
ragged label batches, including an empty utterance.
Run in the build container only:   python tests/golden/make_golden_targets.py   ->  tests/golden/golden_targets_v1.pt"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from espnet.nets.pytorch_backend.transformer.add_sos_eos import add_sos_eos  # noqa: E402
from espnet.nets.pytorch_backend.transformer.mask import target_mask  # noqa: E402


def case(seed, B, L, odim):
    g = torch.Generator().manual_seed(9100 + seed)
    sos = eos = odim - 1
    ys = torch.randint(1, odim - 1, (B, L), generator=g)
    lens = torch.randint(0, L + 1, (B,), generator=g)
    lens[0] = L
    if B > 2:
        lens[2] = 0
    for b in range(B):
        ys[b, int(lens[b]):] = -1
    ys_in, ys_out = add_sos_eos(ys, sos, eos, -1)
    mask = target_mask(ys_in, -1)
    return dict(ys_pad=ys, sos=sos, eos=eos, ys_in=ys_in, ys_out=ys_out, mask=mask, n_tokens=int((ys_out != -1).sum()))


if __name__ == "__main__":
    cases = [case(0, 1, 1, 40), case(1, 4, 9, 40), case(2, 5, 64, 5049), case(3, 3, 130, 5049)]
    torch.save(cases, os.path.join(HERE, "golden_targets_v1.pt"))
    print("wrote", len(cases), "cases;", [tuple(c["ys_in"].shape) for c in cases])
