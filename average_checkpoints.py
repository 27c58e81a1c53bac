"""Checkpoint averaging (reference average_checkpoints.py:6-36): the last ten ``epoch=N.ckpt`` Lightning
checkpoints are averaged key by key (floating tensors /, integer tensors //) into ``model_avg_10.pth`` holding
the bare ``E2E.state_dict()`` (the ``model.`` prefix of the LightningModule is stripped)."""
import os

import torch


def average_checkpoints(last):
    total, n = None, len(last)
    for path in last:
        sd = torch.load(path, map_location="cpu")["state_dict"]
        sd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
        if total is None:
            total = {k: v.clone() for k, v in sd.items()}
        else:
            for k, v in sd.items():
                total[k] += v
    for k, v in total.items():
        if v is None:
            continue
        if v.is_floating_point():
            v /= n
        else:
            v //= n
    return total


def ensemble(args):
    folder = os.path.join(args.exp_dir, args.exp_name)
    last = [os.path.join(folder, f"epoch={n}.ckpt") for n in range(args.max_epochs - 10, args.max_epochs)]
    out = os.path.join(folder, "model_avg_10.pth")
    torch.save(average_checkpoints(last), out)
    return out
