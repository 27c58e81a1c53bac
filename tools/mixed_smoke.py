"""Small-model check of the numerical modes against the fp32 oracle (GPU if present, else the emulator): loss errors, gradient cosines, twin statistics."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
from auto_avsr_amd import _lib, build
pass
from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e import E2E
from oracle import avsr_oracle as O
from synth import synth_state_dict
import test_modules as TM
modality = sys.argv[1] if len(sys.argv) > 1 else "video"
torch.manual_seed(0)
odim = 72
m = TM.no_dropout(E2E(odim, modality, adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2, cnn_module_kernel=7))
sd = synth_state_dict(m.state_dict(), 13)
m.load_state_dict(sd, strict=True)
dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
if dev.type == "cpu":
    _lib._install_for_tests(build.build_emu())
m.to(dev).train()
x, lengths, y = TM.synth_batch(modality, 2, 9, 4, odim, seed=8)
osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
(loss_r, ctc_r, att_r, acc_r), _ = O.e2e_forward(osd, x, lengths, y, modality=modality, heads=2)
loss_r.backward()
for mode in ["bf16", "mixed", "hpf"]:
    AF.invalidate_weight_cache()
    m.load_state_dict(sd, strict=True)
    for p in m.parameters(): p.grad = None
    AF._TWIN_MIN = 0
    AF._twin_stats.update(made=0, used=0, cast=0)
    with AF.numerics(mode):
        loss, loss_ctc, loss_att, acc = m(x.to(dev), lengths.to(dev), y.to(dev))
        loss.backward()
    cos = []
    for k, p in m.named_parameters():
        a, b = p.grad.double().flatten().cpu(), osd[k].grad.double().flatten()
        if b.norm() > 1e-4 * max(1.0, float(osd[k].double().norm())):
            cos.append((float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), k))
    print(mode, "ctc rel", abs(float(loss_ctc) - float(ctc_r)) / abs(float(ctc_r)), "att rel", abs(float(loss_att) - float(att_r)) / abs(float(att_r)),
          "min cos", min(cos), AF._twin_stats, AF.mode())
