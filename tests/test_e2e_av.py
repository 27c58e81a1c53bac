"""Audio-visual composition (auto_avsr_amd/e2e_av.py; no counterpart in the reference snapshot) against the
composition of the reference-pinned oracle parts (oracle/avsr_oracle.py: e2e_av_forward), precise mode, dropout off.
Runs on the emulator build in the CPU suite and on the MI355X in the GPU suite."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
from synth import synth_batch, synth_state_dict  # noqa: E402

import avsr_oracle as O  # noqa: E402
from auto_avsr_amd import _lib  # noqa: E402
from auto_avsr_amd import functional as AF  # noqa: E402
from auto_avsr_amd.e2e_av import E2EAV  # noqa: E402


def test_e2e_av_small_vs_oracle_composition(dev):
    was_precise = AF._state["precise"]
    AF.set_precise(True)
    AF.invalidate_weight_cache()
    try:
        torch.manual_seed(0)
        odim = 48
        m = E2EAV(odim, adim=128, aheads=2, eunits=256, elayers=1, dunits=256, dlayers=1, cnn_module_kernel=7, fusion_hdim=320)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        sd = synth_state_dict(m.state_dict(), 17)
        m.load_state_dict(sd, strict=True)
        m.to(dev).train()
        video, lengths, y = synth_batch("video", 2, 6, 3, odim, seed=4)
        audio, _, _ = synth_batch("audio", 2, 6, 3, odim, seed=5)
        osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running_" not in k else v.clone())
               for k, v in sd.items()}
        (loss_r, ctc_r, att_r, acc_r), _ = O.e2e_av_forward(osd, video, audio, lengths, y, heads=2)
        loss_r.backward()
        loss, loss_ctc, loss_att, acc = m(video.to(dev), audio.to(dev), lengths.to(dev), y.to(dev))
        loss.backward()
        assert abs(float(loss_ctc) - float(ctc_r)) < 1e-3 * abs(float(ctc_r))
        assert abs(float(loss_att) - float(att_r)) < 1e-3 * abs(float(att_r))
        assert abs(acc - acc_r) < 1e-6
        ref_norm = {k: float(osd[k].grad.double().norm()) for k, _ in m.named_parameters()}
        top = max(ref_norm.values())
        bad = []
        for k, p in m.named_parameters():
            got = float(p.grad.double().norm())
            if abs(got - ref_norm[k]) > 1e-2 * ref_norm[k] + 1e-4 * top:
                bad.append((k, got, ref_norm[k]))
        assert not bad, bad[:6]
        # both stacks and the fusion head receive gradient
        for name in ("frontend.trunk", "aux_frontend.trunk", "encoder.encoders.0", "aux_encoder.encoders.0", "fusion.fc1", "fusion.fc2"):
            assert any(k.startswith(name) and float(p.grad.abs().sum()) > 0 for k, p in m.named_parameters()), name
    finally:
        AF.invalidate_weight_cache()
        AF.set_precise(was_precise)
