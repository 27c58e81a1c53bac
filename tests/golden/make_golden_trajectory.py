"""Training-trajectory goldens (round 5; VERDICT r4 "what's missing" 5): the REFERENCE's training loop -- its E2E modules (small
instance, SURVEY F3 replacement as in make_golden_config0.py; the full-size ResNet-18 front-end trains too), torch.optim.AdamW(lr
1e-3, betas (0.9, 0.98), weight decay 0.03) + WarmupCosineScheduler stepped per batch + clip_grad_norm_(10)
(lightning.py:48-52,86-114, train.py:41, cosine.py) -- on four small synthetic batches cycled:

* "nodrop": 50 steps with every dropout probability 0 -> per-step (loss, loss_ctc, loss_att, acc, gradient norm before clipping);
  the product's loop (FusedAdamW, hpf / mixed numerics) must follow it step by step;
* "drop":  200 steps with the reference's dropout rates (0.1 at 63 sites) under torch's RNG -> per-step losses; the product draws
  its masks from its own counter-based generator, so only statistics are comparable (mean loss over windows of the run).

Build container only:   python tests/golden/make_golden_trajectory.py   ->  tests/golden/golden_trajectory_v1.pt"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
from synth import synth_state_dict  # noqa: E402

import math  # noqa: E402

from espnet.nets.pytorch_backend.ctc import CTC  # noqa: E402
from espnet.nets.pytorch_backend.decoder.transformer_decoder import TransformerDecoder  # noqa: E402
from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E  # noqa: E402
from espnet.nets.pytorch_backend.encoder.conformer_encoder import ConformerEncoder  # noqa: E402

from trajectory_common import D, FIXTURE, H, NDEC, NENC, ODIM, SEED, U, WARMUP, TOTAL, batch  # noqa: E402


def build(dropout):
    torch.manual_seed(0)
    m = E2E(ODIM, "video")
    m.proj_encoder = torch.nn.Linear(512, D)
    m.encoder = ConformerEncoder(attention_dim=D, attention_heads=H, linear_units=U, num_blocks=NENC)
    m.decoder = TransformerDecoder(odim=ODIM, attention_dim=D, attention_heads=H, linear_units=U, num_blocks=NDEC)
    m.ctc = CTC(ODIM, D, 0.1, reduce=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), SEED))
    if not dropout:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    return m.train()


def run(steps, dropout):
    m = build(dropout)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03)
    # The reference's WarmupCosineScheduler (cosine.py:6-25) cannot be constructed on this torch (2.10 dropped the `verbose`
    # argument it forwards, cosine.py:18); its get_lr() (cosine.py:20-25) is restated as a LambdaLR -- `_step_count` there is
    # `last_epoch + 1` here.  warmup_epochs = WARMUP, total_epochs = TOTAL, steps_per_epoch = 1: the schedule moves per batch.
    def factor(last_epoch):
        n = last_epoch + 1
        if n < WARMUP:
            return n / WARMUP
        return 0.5 * (1 + math.cos(math.pi * (n - WARMUP) / (TOTAL - WARMUP)))

    sched = torch.optim.lr_scheduler.LambdaLR(opt, factor)
    torch.manual_seed(1234)
    rows = []
    t0 = time.time()
    for s in range(steps):
        x, lens, y = batch(s % 4)
        loss, loss_ctc, loss_att, acc = m(x, lens, y)
        opt.zero_grad()
        loss.backward()
        gn = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 10.0))
        lr = opt.param_groups[0]["lr"]
        opt.step()
        sched.step()
        rows.append(dict(loss=float(loss), loss_ctc=float(loss_ctc), loss_att=float(loss_att), acc=float(acc), grad_norm=gn, lr=lr))
        if s % 10 == 0 or s == steps - 1:
            print(("drop" if dropout else "nodrop"), s, rows[-1], f"{time.time() - t0:.0f}s", flush=True)
    probe = {k: v.detach().flatten()[:16].clone() for k, v in m.state_dict().items()
             if k in ("proj_encoder.weight", "encoder.encoders.0.feed_forward.w_1.weight", "decoder.output_layer.weight",
                      "frontend.trunk.layer1.0.conv1.weight", "encoder.encoders.1.conv_module.norm.running_var")}
    return dict(steps=rows, probe=probe)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    out = dict(torch_version=torch.__version__, nodrop=run(50, False), drop=run(200, True))
    torch.save(out, FIXTURE)
