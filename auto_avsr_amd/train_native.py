"""Native training driver: one process per GPU (torchrun), torch.distributed over RCCL, no Lightning.

Implements what the reference gets from `Trainer(sync_batchnorm=True, DDPStrategy(find_unused_parameters=False),
gradient_clip_val=10.0, callbacks=[ModelCheckpoint(save_top_k=10, save_last=True)])` + `ModelModule.training_step /
validation_step / configure_optimizers` + `ensemble(args)` (train.py:17-50, lightning.py:48-52,86-114,
average_checkpoints.py): DDP gradient averaging, cross-rank BatchNorm statistics, the W / sum(B) loss rescale, global-norm
clipping at 10, AdamW(0.9, 0.98) with the per-step warm-up cosine schedule, a validation pass and an `epoch=N.ckpt`
(Lightning layout: state_dict keys prefixed `model.`) per epoch with the ten newest kept, `last.ckpt` carrying the
optimizer state for `--ckpt-path` resume, and the 10-checkpoint average `model_avg_10.pth` at the end -- on the synthetic
LRS3-shaped workload (no dataset on the box)."""
import os
import time

import torch
import torch.distributed as dist


class _Hot(torch.nn.Module):
    """forward_tensors() behind nn.Module.__call__ so that DDP's reducer hooks see the step."""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, x, lens, y):
        return self.m.forward_tensors(x, lens, y)


def save_checkpoint(folder, epoch, model, opt, global_step, keep=10, opt_state=None):
    """rank 0: `epoch=N.ckpt` in the layout average_checkpoints.py / lightning.py:44 read ({"state_dict": {"model.<key>"}}),
    the `keep` newest kept (ModelCheckpoint(monitor="monitoring_step", mode="max", save_top_k=10): the monitored value
    is the global step, so "top 10" = the ten latest), plus last.ckpt with the optimizer state."""
    os.makedirs(folder, exist_ok=True)
    sd = {"model." + k: v.detach().cpu() for k, v in model.state_dict().items()}
    meta = {"epoch": epoch, "global_step": global_step}
    torch.save({"state_dict": sd, **meta}, os.path.join(folder, f"epoch={epoch}.ckpt"))
    # (opt_state: the optimizer state gathered beforehand by ALL ranks -- a sharded optimizer's state_dict() is a collective)
    torch.save({"state_dict": sd, "optimizer": opt_state if opt_state is not None else opt.state_dict(), **meta},
               os.path.join(folder, "last.ckpt"))
    old = os.path.join(folder, f"epoch={epoch - keep}.ckpt")
    if os.path.exists(old):
        os.remove(old)


def load_checkpoint(path, model, opt):
    """--ckpt-path resume (train.py:49 `trainer.fit(..., ckpt_path=...)`): weights, optimizer state, position."""
    from . import functional as AF

    ck = torch.load(path, map_location="cpu")
    model.load_state_dict({k[len("model."):]: v for k, v in ck["state_dict"].items() if k.startswith("model.")})
    global_step = int(ck.get("global_step", 0))
    if "optimizer" in ck:
        opt.load_state_dict(ck["optimizer"])
    else:
        # a weights-only file (an `epoch=N.ckpt` written without optimizer state, a Lightning checkpoint whose
        # `optimizer_states` use torch's layout): the moments restart from zero, but the step counter -- which drives the
        # warm-up / cosine schedule and Adam's bias correction -- must not: continue it from the stored global step
        import warnings

        warnings.warn(f"{path}: no native optimizer state; resuming with zero Adam moments at step {global_step} of the schedule")
        opt.state[0] = float(global_step)
    AF.invalidate_weight_cache()  # every cached bf16 copy belongs to the old weights
    return int(ck.get("epoch", -1)) + 1, global_step


@torch.no_grad()
def validate(model, batches, world):
    """ModelModule.validation_step over the val batches (lightning.py:95-96,104-113): eval-mode forward, metrics averaged
    over batches and ranks."""
    was_training = model.training
    model.eval()
    tot = torch.zeros(5, device=batches[0][0].device) if batches else None
    for x, lens, y, _ in batches:
        loss, loss_ctc, loss_att, hits, ntok = model.forward_tensors(x, lens, y)
        tot += torch.stack([loss, loss_ctc, loss_att, hits / ntok.clamp_min(1), torch.ones_like(loss)])
    model.train(was_training)
    if tot is None:
        return None
    if world > 1:
        dist.all_reduce(tot)
    n = float(tot[4])
    return {"loss_val": float(tot[0]) / n, "loss_ctc_val": float(tot[1]) / n, "loss_att_val": float(tot[2]) / n,
            "decoder_acc_val": float(tot[3]) / n}


class _SyntheticSource:
    """The synthetic LRS3-shaped corpus: length-bucketed batches, a fresh seeded order per epoch and rank."""

    def __init__(self, args, model, dev, rank, world):
        from .synthetic import bucket_batches, utterance_lengths

        self.args, self.odim, self.dev, self.rank, self.world = args, model.odim, dev, rank, world
        self.lengths = utterance_lengths(getattr(args, "synthetic_utterances", 0) or 20000)
        self.all_batches = bucket_batches(self.lengths, args.max_frames, args.train_num_buckets)
        self.steps_per_epoch = (len(self.all_batches) + world - 1) // world

    def epoch(self, epoch, global_step):
        from .synthetic import make_batch, rank_batches

        batches = rank_batches(self.all_batches, self.rank, self.world, seed=epoch)
        assert len(batches) == self.steps_per_epoch
        for i, idxs in enumerate(batches):
            x, lens, y, _ = make_batch(self.lengths, idxs, self.args.modality, self.odim, seed=global_step + i, device=self.dev,
                                       on_device=True)
            yield x, lens, y

    def val_batches(self, n):
        from .synthetic import bucket_batches, make_batch, rank_batches, utterance_lengths

        val_lengths = utterance_lengths(2000, seed=43)
        val_all = bucket_batches(val_lengths, 1000, 1)  # val_dataloader: max_frames 1000, one bucket (data_module.py:156-158)
        return [make_batch(val_lengths, b, self.args.modality, self.odim, seed=10_000 + i, device=self.dev)
                for i, b in enumerate(rank_batches(val_all, self.rank, self.world, seed=1)[:n])]


class _FileSource:
    """File-backed data through the reference's DataModule surface (datamodule/data_module.py: AVDataset + CustomBucketDataset +
    the device-side transform / collation of DeviceBatches), sharded across ranks by the DistributedSampler DataModule installs."""

    def __init__(self, args, model, dev, rank, world):
        from datamodule.data_module import DataModule

        self.dm = DataModule(args, train_num_buckets=args.train_num_buckets, device=str(dev),
                             num_workers=int(getattr(args, "num_workers", 10)))
        self.dev = dev
        self.steps_per_epoch = len(self.dm.train_dataloader())

    def _triples(self, loader):
        for b in loader:
            yield b["inputs"].to(self.dev), b["input_lengths"].to(self.dev), b["targets"].to(self.dev)

    def epoch(self, epoch, global_step):
        self.dm._train_loaders = epoch - 1  # (DataModule._train_epoch: the loader handed out next serves this epoch)
        return self._triples(self.dm.train_dataloader())

    def val_batches(self, n):
        import itertools

        return [(x, l, y, None) for x, l, y in itertools.islice(self._triples(self.dm.val_dataloader()), n)] if n else []


def _batch_source(args, model, dev, rank, world):
    if getattr(args, "synthetic", False) or not getattr(args, "train_file", None):
        return _SyntheticSource(args, model, dev, rank, world)
    return _FileSource(args, model, dev, rank, world)


def fit(model, args, dev, rank=0, world=1, backend="nccl", log=print):
    """The training loop proper, on an already constructed `E2E`-like model (tests run it on a small instance over
    gloo + the emulator).  Returns the list of per-step losses of this rank.  Whatever the loop attached to the model or the
    process -- gradient-bucket hooks on the parameters, RCCL communicators, the cross-rank BatchNorm switch -- is released on
    the way out, also when a step raises: a second fit() on the same model (fine-tuning after pre-training in one process)
    starts clean instead of running two hook sets on stale communicators."""
    from . import functional as AF

    held = {"native": None, "stepper": None}
    old_mode = AF._save_mode()
    # The whole loop runs on a stream of its own, never on the legacy default stream: gradient-accumulation nodes remember the
    # stream they were created on, and a hipGraph capture must not meet work bound to the default stream (it would have to
    # synchronise with it, which is illegal under capture) -- the same rule bench.py's warm-up steps follow.
    work = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
    if work is not None:
        work.wait_stream(torch.cuda.current_stream(dev))
    try:
        if work is None:
            return _fit(model, args, dev, rank, world, backend, log, held)
        with torch.cuda.stream(work):
            out = _fit(model, args, dev, rank, world, backend, log, held)
        torch.cuda.current_stream(dev).wait_stream(work)
        return out
    finally:
        AF._restore_mode(old_mode)
        fit.last_stats = dict(held["stepper"].stats, tail_ms=held.get("tail_ms")) if held["stepper"] is not None else None
        if held["native"] is not None:
            held["native"].close()
        AF.set_bn_sync(None)


def _max_graphs():
    """hipGraphs (batch shapes) the loop keeps: 384 distinct (B, T, L) shapes at max-frames 1600 on the synthetic corpus; one
    graph holds ~50 MB of static inputs, the activations' pool is shared."""
    return int(os.environ.get("AVSR_MAX_GRAPHS", "512"))


class NativeStepper:
    """ONE training step as one callable: forward + backward + (data-parallel exchange, cross-rank BatchNorm, W / sum(B)) + fused
    global-norm clip / AdamW / warm-up cosine + the per-step weight re-casts -- what `bench.py` times -- replayed as a hipGraph
    per batch shape from a shape's second visit on (graph_step.StepGraphs).  Two drivers: `fit()` below (train.py's native
    loop) and `lightning.ModelModule` in its manual-optimisation mode (`--trainer-step native`: a Lightning `Trainer` then only
    feeds batches and runs callbacks, lightning.py:86-114 / train.py:30-42), so that the Trainer path runs at the speed of the
    benchmarked step instead of eager launches + torch's foreach AdamW.

    stepper(x, lens, y) -> (loss, loss_ctc, loss_att, n_correct, n_tokens) as device scalars (static tensors of the graph when
    replayed: read them before the next step of the same shape).  close() removes the hooks / communicators it installed."""

    def __init__(self, model, args, dev, rank, world, steps_per_epoch, log=print, own_stream=True):
        from . import functional as AF
        from .graph_step import StepGraphs
        from .optim import FusedAdamW

        self.model, self.args, self.dev, self.rank, self.world = model, args, dev, rank, world
        self.comms, self.buckets, self.shard = [], None, False
        # own_stream: run every step on a stream of this object, joined to the caller's stream before and after.  A hipGraph capture
        # must never meet work bound to the legacy default stream (gradient-accumulation nodes remember the stream they were
        # created on), and a Lightning Trainer calls training_step on the default stream; fit() below already runs its whole loop
        # on a side stream and passes False
        self._work = torch.cuda.Stream(device=dev) if (own_stream and dev.type == "cuda") else None
        # train.py --deterministic / AVSR_DETERMINISTIC=1: every gradient sum in a fixed order (functional.set_deterministic)
        self._det_before = AF.deterministic()
        if getattr(args, "deterministic", False) or os.environ.get("AVSR_DETERMINISTIC", "0") == "1":
            AF.set_deterministic(True)
        AF.manual_seed(42 + rank)
        self.seed_dev = seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        AF.set_seed_tensor(seed_dev)
        hot = _Hot(model)
        buckets = None
        if world > 1:
            # train.py:31,37: cross-rank BatchNorm (the kernels' own statistics exchange) + DDP gradient averaging
            AF.set_bn_sync(dist.group.WORLD)
            if os.environ.get("AVSR_DDP", "torch") == "buckets":
                # this build's own bucketed RCCL all-reduce (ddp.GradBuckets); on GPUs every collective of the step -- gradient
                # buckets, BatchNorm statistics -- goes straight to RCCL's C API (comm.StreamComm: a ctypes call instead of a
                # c10d Work object per collective; 64 BatchNorm collectives per step make that the larger share of the host time)
                from .ddp import GradBuckets

                comm_grads = None
                if dev.type == "cuda":
                    from .comm import StreamComm

                    comm_bn, comm_grads = StreamComm.from_process_group(), StreamComm.from_process_group()
                    self.comms += [comm_bn, comm_grads]
                    AF.set_bn_sync(dist.group.WORLD, comm=comm_bn)
                # wire format of the gradient buckets: f32 like the reference's DDP all-reduce unless asked otherwise
                # (--grad-wire bf16 / AVSR_GRAD_WIRE=bf16: half the bytes per xGMI link, bf16 sums across the ranks)
                wire = getattr(args, "grad_wire", None) or os.environ.get("AVSR_GRAD_WIRE") or "f32"
                # AVSR_SHARD_OPT=1: reduce-scatter instead of all-reduce, the optimizer on this rank's 1 / world slice of every
                # bucket, all-gather of the updated flat parameter buffers (optim.ShardedAdamW; opt-in, see there)
                self.shard = os.environ.get("AVSR_SHARD_OPT", "0") == "1"
                buckets = self.buckets = GradBuckets(model.parameters(), group=dist.group.WORLD, comm=comm_grads, wire=wire,
                                                     spare=_max_graphs() + 8, shard=self.shard)
                if self.shard:
                    AF.invalidate_weight_cache()  # (the parameters moved into the flat buffers: cached copies are keyed by address)
            else:
                hot = torch.nn.parallel.DistributedDataParallel(
                    hot, device_ids=[dev.index] if dev.type == "cuda" else None, find_unused_parameters=False,
                    broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=64)
        # lightning.py:48-52 + train.py:41 + cosine.py as one fused multi-tensor step (optim.py): AdamW(.9/.98), clip 10,
        # per-step warm-up cosine; step count / lr / gradient norm stay on the device
        if buckets is not None and buckets.shard:
            from .optim import ShardedAdamW

            opt = self.opt = ShardedAdamW(buckets, lr=args.lr, betas=(0.9, 0.98), weight_decay=args.weight_decay, max_grad_norm=10.0,
                                          warmup_steps=int(args.warmup_epochs * steps_per_epoch),
                                          total_steps=int(args.max_epochs * steps_per_epoch))
        else:
            opt = self.opt = FusedAdamW(model.parameters(), lr=args.lr, betas=(0.9, 0.98), weight_decay=args.weight_decay,
                                        max_grad_norm=10.0, warmup_steps=int(args.warmup_epochs * steps_per_epoch),
                                        total_steps=int(args.max_epochs * steps_per_epoch), cast_weights=dev.type == "cuda",
                                        graph_shapes=_max_graphs())
        params = list(model.parameters())

        def full_step(x, lens, y):
            """ONE training step, start to end, with no host decision that depends on the batch's values: what runs eagerly the
            first time a batch shape shows up and what a hipGraph of that shape replays afterwards (graph_step.StepGraphs)."""
            for p in params:  # (gradients of a replayed step live in the graph's pool: never accumulate into them)
                p.grad = None
            seed_dev.add_(1)
            AF.manual_seed(42 + rank)  # restart the per-site counter: mask = f(rank, site index, seed_dev = global step)
            if buckets is not None:
                buckets.begin_step()
            AF.new_step()
            AF.refresh_weight_cache()  # conv-weight permutes; the Linear copies were rewritten by the optimizer step itself
            loss, loss_ctc, loss_att, hits, ntok = hot(x, lens, y)
            if world > 1:
                bs = torch.full((1,), float(x.shape[0]), device=dev)
                allb = torch.empty(world, device=dev)
                if buckets is not None and buckets.comm is not None:
                    AF._state["bn_comm"].all_gather(allb, bs)
                else:
                    dist.all_gather_into_tensor(allb, bs)
                loss = loss * (world / allb.sum())  # lightning.py:88-90
            loss.backward()
            if buckets is not None:
                buckets.finish()
            opt.step()
            if buckets is not None and not buckets.rebuilt and not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
                for p in params:  # (after the first step: buckets in the order the gradients arrived -- ddp.GradBuckets.rebuild_by_arrival)
                    p.grad = None
                buckets.rebuild_by_arrival()
            return loss.detach(), loss_ctc.detach(), loss_att.detach(), hits, ntok

        # hipGraph replay per batch shape (what bench.py times): single-rank runs, and data-parallel runs whose collectives are all
        # stream operations on RCCL's C API (AVSR_DDP=buckets on GPUs); torch's DDP reducer cannot be captured
        graph_ok = dev.type == "cuda" and not getattr(args, "no_graph", False) and \
            (world == 1 or (buckets is not None and buckets.comm is not None))

        def capture_failed(e):
            # a capture that dies mid-backward leaves partial bucket counts, pending reductions and gradients that live in the dead
            # graph's pool: reset all of it, or the eager retry never flushes those buckets and finish() raises on this rank while
            # the others wait in a collective (round-5 advisor finding)
            log(f"[rank {rank}] hipGraph capture failed ({type(e).__name__}: {str(e)[:160]}); eager from here on")
            for p in params:
                p.grad = None
            if buckets is not None:
                buckets.abort_step()

        def released(key):  # a graph is gone: its pinned pointer tables go back to their owners
            opt.release_captured(key)
            if buckets is not None:
                buckets.release_captured(key)

        # capacity: the first AVSR_MAX_GRAPHS (default 512) shapes that show up twice are captured and kept for the run -- no
        # eviction (graph_step.py); the optimizer's and the buckets' pre-pinned tables are sized for exactly that many captures
        self.stepper = StepGraphs(full_step, enabled=graph_ok, capture_after=1, thread_local=world > 1, max_graphs=_max_graphs(),
                                  on_fail=capture_failed, on_evict=released)

    @property
    def stats(self):
        return self.stepper.stats

    def __call__(self, x, lens, y):
        if self._work is None:
            return self.stepper(x, lens, y)
        cur = torch.cuda.current_stream(self.dev)
        self._work.wait_stream(cur)
        with torch.cuda.stream(self._work):
            out = self.stepper(x, lens, y)
        cur.wait_stream(self._work)
        return out

    def close(self):
        from . import functional as AF

        if self.buckets is not None:
            self.buckets.remove()
            self.buckets = None
        for c in self.comms:
            c.close()
        self.comms = []
        AF.set_bn_sync(None)
        if AF._state.get("seed_dev") is self.seed_dev:
            AF.set_seed_tensor(None)  # (the per-step dropout counter of THIS loop: not the next caller's)
        if AF.deterministic() != self._det_before:
            AF.set_deterministic(self._det_before)


def _fit(model, args, dev, rank, world, backend, log, held):
    from . import functional as AF
    source = _batch_source(args, model, dev, rank, world)
    # every rank sees the same number of batches per epoch (DistributedSampler pads): the schedule lengths below and the
    # number of collectives per epoch are identical on all ranks
    steps_per_epoch = source.steps_per_epoch
    ns = held["native"] = NativeStepper(model, args, dev, rank, world, steps_per_epoch, log=log, own_stream=False)
    opt, seed_dev, stepper = ns.opt, ns.seed_dev, ns.stepper
    held["stepper"] = stepper
    folder = os.path.join(args.exp_dir, args.exp_name) if getattr(args, "exp_dir", None) else None
    start_epoch, global_step = 0, 0
    if getattr(args, "ckpt_path", None):
        start_epoch, global_step = load_checkpoint(args.ckpt_path, model, opt)
        seed_dev.add_(global_step)  # dropout masks are a function of (rank, global step, site): a resumed run continues them
    val = source.val_batches(getattr(args, "val_batches", 0) or 0)
    max_steps = getattr(args, "steps", None)
    losses = []
    t0 = time.time()
    done = False
    mode = getattr(args, "numerics", None)  # train.py: --numerics, default "mixed" (what bench.py times); None: the caller's mode
    tail, t_tail = int(getattr(args, "time_last", 0) or 0), None
    if mode is not None:
        AF.set_mode(mode)
    for epoch in range(start_epoch, args.max_epochs):
        # reload_dataloaders_every_n_epochs=1 + shuffle=True (train.py:39, data_module.py:140): a new order every epoch
        nb = 0
        for bi, (x, lens, y) in enumerate(source.epoch(epoch, global_step)):
            nb += 1
            if tail and max_steps and global_step == max_steps - tail:  # --time-last N: wall clock of the last N steps
                torch.cuda.synchronize() if dev.type == "cuda" else None
                t_tail = time.perf_counter()
            loss, loss_ctc, loss_att, hits, ntok = stepper(x, lens, y)
            global_step += 1
            if t_tail is not None and global_step == max_steps:
                torch.cuda.synchronize() if dev.type == "cuda" else None
                held["tail_ms"] = (time.perf_counter() - t_tail) / tail * 1e3
                log(f"last {tail} steps: {held['tail_ms']:.2f} ms / step ({stepper.stats})")
            every = getattr(args, "log_every", 10)
            if every and ((global_step - 1) % every == 0 or global_step == max_steps):
                losses.append(float(loss.detach()))
                if rank == 0:
                    log(f"epoch {epoch} step {global_step} loss {losses[-1]:.4f} ctc {float(loss_ctc.detach()):.4f} att "
                        f"{float(loss_att.detach()):.4f} acc {float(hits.detach()) / max(float(ntok.detach()), 1):.4f} lr {opt.last_lr:.2e} gnorm "
                        f"{opt.last_grad_norm:.2f} ({time.time() - t0:.1f}s)")
            if max_steps and global_step >= max_steps:
                done = True
                break
        if done and nb < steps_per_epoch:
            break  # stopped inside an epoch (--steps): no end-of-epoch work
        metrics = validate(model, val, world)
        if rank == 0 and metrics:
            log(f"epoch {epoch} validation: " + " ".join(f"{k} {v:.4f}" for k, v in metrics.items()))
        opt_state = opt.state_dict() if (folder and getattr(opt, "collective_state", False)) else None  # (every rank: a collective)
        if rank == 0 and folder:
            save_checkpoint(folder, epoch, model, opt, global_step, opt_state=opt_state)
        if world > 1:
            dist.barrier()
        if done:
            break
    else:
        if rank == 0 and folder and args.max_epochs >= 10:
            from average_checkpoints import ensemble

            log(f"averaged checkpoint: {ensemble(args)}")
    return losses


def run(args):
    from lightning import ModelModule

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(42)
    module = ModelModule(args).to(dev).train()
    fit(module.model, args, dev, rank, world)
    if world > 1:
        dist.destroy_process_group()
