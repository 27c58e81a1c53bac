"""Native training driver: one process per GPU (torchrun), torch.distributed over RCCL, no Lightning.

Implements what the reference gets from `Trainer(sync_batchnorm=True, DDPStrategy(find_unused_parameters=False),
gradient_clip_val=10.0)` + `ModelModule.training_step` + `configure_optimizers` (train.py:30-42,
lightning.py:48-52,86-94): DDP gradient averaging, cross-rank BatchNorm statistics, the W / sum(B) loss rescale,
global-norm clipping at 10, AdamW(0.9, 0.98) and the per-step warm-up cosine schedule, on the synthetic
LRS3-shaped workload (no dataset on the box)."""
import os
import time

import torch
import torch.distributed as dist


def run(args):
    from lightning import ModelModule

    from . import functional as AF
    from .synthetic import bucket_batches, make_batch, rank_batches, utterance_lengths

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
        AF.set_bn_sync(dist.group.WORLD)
    torch.manual_seed(42)
    module = ModelModule(args).to(dev).train()
    model = module.model
    AF.manual_seed(42 + rank)
    seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    AF.set_seed_tensor(seed_dev)

    class Hot(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x, lens, y):
            return self.m.forward_tensors(x, lens, y)

    hot = Hot(model)
    if world > 1:
        hot = torch.nn.parallel.DistributedDataParallel(hot, device_ids=[local], find_unused_parameters=False,
                                                        broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=64)
    lengths = utterance_lengths()
    batches = rank_batches(bucket_batches(lengths, args.max_frames, args.train_num_buckets), rank, world, seed=0)
    # lightning.py:48-52 + train.py:41 + cosine.py as one fused multi-tensor step (optim.py): AdamW(.9/.98), clip 10,
    # per-step warm-up cosine; step count / lr / gradient norm stay on the device
    from .optim import FusedAdamW

    opt = FusedAdamW(model.parameters(), lr=args.lr, betas=(0.9, 0.98), weight_decay=args.weight_decay, max_grad_norm=10.0,
                     warmup_steps=args.warmup_epochs * len(batches), total_steps=args.max_epochs * len(batches),
                     cast_weights=True)
    total = args.steps or args.max_epochs * len(batches)
    t0 = time.time()
    for step in range(total):
        x, lens, y, frames = make_batch(lengths, batches[step % len(batches)], args.modality, model.odim, seed=step,
                                        device=dev)
        seed_dev.add_(1)
        AF.new_step()
        AF.refresh_weight_cache()  # conv-weight permutes; the Linear copies were rewritten by the optimizer step itself
        loss, loss_ctc, loss_att, hits, ntok = hot(x, lens, y)
        if world > 1:
            bs = torch.tensor([float(x.shape[0])], device=dev)
            allb = torch.empty(world, device=dev)
            dist.all_gather_into_tensor(allb, bs)
            loss = loss * (world / allb.sum())
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if rank == 0 and (step % 10 == 0 or step == total - 1):
            print(f"step {step} loss {float(loss):.4f} ctc {float(loss_ctc):.4f} att {float(loss_att):.4f} "
                  f"acc {float(hits) / max(float(ntok), 1):.4f} lr {opt.last_lr:.2e} gnorm {opt.last_grad_norm:.2f} "
                  f"({time.time() - t0:.1f}s)", flush=True)
    if world > 1:
        dist.destroy_process_group()
