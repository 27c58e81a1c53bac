"""Diagnostic: which ingredient of the single-rank RCCL run changes the first-step gradients? (plain twice / SyncBN only /
DDP only / both), precise mode, full or small model."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, torch.distributed as dist
from synth import synth_batch, synth_state_dict
from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e import E2E
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
full = "--full" in sys.argv
V = 5049 if full else 41
def build():
    torch.manual_seed(0)
    m = E2E(5049, "video") if full else E2E(41, "video", adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2, cnn_module_kernel=7)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    m.load_state_dict(synth_state_dict(m.state_dict(), 31))
    return m.to(dev).train()
class Hot(torch.nn.Module):
    def __init__(self, mm): super().__init__(); self.m = mm
    def forward(self, x, lens, y): return self.m.forward_tensors(x, lens, y)[0]
if "--nopad" in sys.argv:
    x, lens, y = (t.to(dev) for t in synth_batch("video", 3, 24, 3, V, seed=12, lengths=[24, 24, 24]))
    y = y.clamp_min(1)
else:
    x, lens, y = (t.to(dev) for t in synth_batch("video", 3, 24 if full else 8, 3, V, seed=12, lengths=[24, 20, 17] if full else [8, 6, 5]))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29433")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
def run(ddp, bnsync, view=True, eps_scale=1.0):
    AF.invalidate_weight_cache(); AF.set_precise(True); AF.set_bn_sync(dist.group.WORLD if bnsync else None)
    m = build(); hot = Hot(m)
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm): mod.eps = mod.eps * eps_scale
    if ddp:
        hot = torch.nn.parallel.DistributedDataParallel(hot, device_ids=[0], find_unused_parameters=False, broadcast_buffers=False, gradient_as_bucket_view=view, bucket_cap_mb=64)
    AF.new_step(); AF.refresh_weight_cache()
    loss = hot(x, lens, y); loss.backward(); torch.cuda.synchronize()
    g = {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}
    AF.set_bn_sync(None); AF.set_precise(False)
    return float(loss.detach()), g
def cmp(a, b):
    gmax = max(float(v.norm()) for v in a.values())
    errs = {k: float((b[k] - a[k]).norm()) / max(float(a[k].norm()), 1e-3 * gmax) for k in a}
    top = sorted(errs, key=errs.get, reverse=True)[:12]
    return [(k, round(errs[k], 6)) for k in top]
l0, g0 = run(False, False)
def poison(v):
    # fill the caching allocator's free blocks with a marker: a kernel reading outside its tensors picks it up
    blocks = [torch.full((n,), v, device=dev) for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16) for _ in range(3)]
    del blocks
for name, args in (("plain, BN eps * (1 + 1e-6)", (False, False, True, 1.000001)), ("bnsync only", (False, True)), ("both", (True, True))):
    if "poisoned" in name: poison(1e30)
    keep = torch.empty(12345, device=dev) if "shifted" in name else None
    l, g = run(*args)
    print(name, "loss diff", abs(l - l0) / abs(l0), "worst grad", cmp(g0, g), flush=True)
