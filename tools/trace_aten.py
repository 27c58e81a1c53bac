"""Which torch (aten) operators -- i.e. launches that are NOT ours -- does one full-size training step issue, and from
where?  Prints (count, operator, call site, first tensor shape) sorted by count.  GPU box:  python tools/trace_aten.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.getcwd())
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e import E2E
from auto_avsr_amd.optim import FusedAdamW
from auto_avsr_amd.synthetic import bucket_batches, make_batch, rank_batches, utterance_lengths

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = E2E(5049, "video").to(dev).train()
AF.set_precise(False)
AF.manual_seed(1234)
seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
AF.set_seed_tensor(seed_dev)
opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0,
                 warmup_steps=5000, total_steps=75000)
lengths = utterance_lengths()
batches = rank_batches(bucket_batches(lengths, 1600, 400), 0, 1, seed=0)
x, lens, y, _ = make_batch(lengths, batches[len(batches) // 2], "video", 5049, seed=0, device=dev)


def step():
    AF.new_step()
    seed_dev.add_(1)
    AF.refresh_weight_cache()
    loss = model.forward_tensors(x, lens, y)[0]
    loss.backward()
    opt.step()
    model.zero_grad(set_to_none=True)


step()
cnt = collections.Counter()
SKIP = ("view", "as_strided", "detach", "slice", "select", "empty", "unsqueeze", "reshape", "alias", "narrow",
        "transpose", "expand", "permute", "squeeze", "t.default", "_unsafe_view", "resize_", "split", "unbind", "stride",
        "sym_", "is_", "size", "numel", "dim", "storage_offset", "_local_scalar")


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            st = traceback.extract_stack()
            fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in st if "auto_avsr_amd" in f.filename][-2:]
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            cnt[(name, tuple(fr), shp if len(fr) == 0 else None)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    step()
torch.cuda.synchronize()
tot = collections.Counter()
for (name, fr, shp), v in cnt.items():
    tot[name] += v
print("totals:", tot.most_common(30))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(v, k)
