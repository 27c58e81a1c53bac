"""Generate the golden vectors under tests/golden/ by running the REFERENCE implementation
(/root/reference, PyTorch CPU).  Run in the build container only:

    python tests/golden/make_golden.py

The fixtures hold only seeds, scalar results and small tensors; weights are regenerated from
tests/golden/synth.py on both sides.  Dropout probabilities are forced to 0 so the numbers are
deterministic; BatchNorm stays in training mode (batch statistics incl. padding)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
from synth import synth_batch, synth_state_dict  # noqa: E402

from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E  # noqa: E402
from espnet.nets.pytorch_backend.encoder.conformer_encoder import ConformerEncoder  # noqa: E402
from espnet.nets.pytorch_backend.decoder.transformer_decoder import TransformerDecoder  # noqa: E402
from espnet.nets.pytorch_backend.nets_utils import make_non_pad_mask  # noqa: E402
from espnet.nets.pytorch_backend.transformer.mask import target_mask  # noqa: E402


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def grad_norms(m):
    return {k: float(p.grad.double().norm()) for k, p in m.named_parameters() if p.grad is not None}


def e2e_case(modality, B, T, L, seed):
    torch.manual_seed(0)
    m = no_dropout(E2E(5049, modality))
    m.load_state_dict(synth_state_dict(m.state_dict(), seed))
    m.train()
    x, lengths, y = synth_batch(modality, B, T, L, 5049, seed)
    loss, loss_ctc, loss_att, acc = m(x, lengths, y)
    loss.backward()
    with torch.no_grad():
        feats = m.frontend(x)
    return dict(modality=modality, B=B, T=T, L=L, seed=seed, loss=float(loss), loss_ctc=float(loss_ctc),
                loss_att=float(loss_att), acc=float(acc), grad_norms=grad_norms(m),
                feats_sample=feats[:, :, :16].clone(), keys=list(m.state_dict().keys()),
                shapes={k: tuple(v.shape) for k, v in m.state_dict().items()})


def encoder_case(seed):
    torch.manual_seed(0)
    enc = no_dropout(ConformerEncoder(attention_dim=128, attention_heads=2, linear_units=256, num_blocks=2,
                                      cnn_module_kernel=7))
    enc.load_state_dict(synth_state_dict(enc.state_dict(), seed))
    enc.train()
    g = torch.Generator().manual_seed(77 + seed)
    B, T = 3, 21
    x = torch.randn(B, T, 128, generator=g, requires_grad=True)
    lengths = torch.tensor([21, 17, 9])
    mask = make_non_pad_mask(lengths).unsqueeze(-2)
    out, _ = enc(x, mask)
    w = torch.randn(out.shape, generator=g)
    (out * w).sum().backward()
    return dict(seed=seed, x=x.detach(), lengths=lengths, out=out.detach(), w=w, dx=x.grad.clone(),
                grad_norms=grad_norms(enc), shapes={k: tuple(v.shape) for k, v in enc.state_dict().items()})


def decoder_case(seed):
    torch.manual_seed(0)
    odim = 61
    dec = no_dropout(TransformerDecoder(odim=odim, attention_dim=128, attention_heads=2, linear_units=256, num_blocks=2))
    dec.load_state_dict(synth_state_dict(dec.state_dict(), seed))
    dec.train()
    g = torch.Generator().manual_seed(99 + seed)
    B, T, L = 3, 19, 7
    memory = torch.randn(B, T, 128, generator=g, requires_grad=True)
    lengths = torch.tensor([19, 12, 5])
    mmask = make_non_pad_mask(lengths).unsqueeze(-2)
    ys_in = torch.randint(1, odim, (B, L), generator=g)
    ys_mask = target_mask(ys_in, -1)
    out, _ = dec(ys_in, ys_mask, memory, mmask)
    w = torch.randn(out.shape, generator=g)
    (out * w).sum().backward()
    return dict(seed=seed, odim=odim, memory=memory.detach(), lengths=lengths, ys_in=ys_in, out=out.detach(), w=w,
                dmemory=memory.grad.clone(), grad_norms=grad_norms(dec),
                shapes={k: tuple(v.shape) for k, v in dec.state_dict().items()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    out = {
        "torch_version": torch.__version__,
        "encoder_small": encoder_case(1),
        "decoder_small": decoder_case(2),
        "e2e_video": e2e_case("video", 2, 10, 4, 3),
        "e2e_audio": e2e_case("audio", 2, 10, 4, 4),
    }
    torch.save(out, os.path.join(HERE, "golden_v1.pt"))
    for k in ("e2e_video", "e2e_audio"):
        c = out[k]
        print(k, c["loss"], c["loss_ctc"], c["loss_att"], c["acc"])
    print("encoder out abs-mean", out["encoder_small"]["out"].abs().mean().item())
    print("decoder out abs-mean", out["decoder_small"]["out"].abs().mean().item())
