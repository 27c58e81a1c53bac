"""Training entry point with the reference's command line (train.py:52-190).  With pytorch_lightning present the
run goes through Lightning's Trainer exactly as the reference configures it (train.py:17-42: SyncBatchNorm, DDP,
gradient clip 10, top-10 + last checkpoints); otherwise -- as in this image -- it is handed to the native
one-process-per-GPU driver ``auto_avsr_amd.train_native`` (torch.distributed over RCCL)."""
import os
from argparse import ArgumentParser


def parse_args(argv=None):
    p = ArgumentParser()
    p.add_argument("--exp-dir", default="./exp", type=str, help="directory for checkpoints and logs")
    p.add_argument("--exp-name", default="run", type=str)
    p.add_argument("--group-name", default=None, type=str)
    p.add_argument("--modality", default="video", type=str, choices=["audio", "video"])
    p.add_argument("--root-dir", default=None, type=str)
    p.add_argument("--train-file", default=None, type=str)
    p.add_argument("--val-file", default="lrs3_test_transcript_lengths_seg16s.csv", type=str)
    p.add_argument("--test-file", default="lrs3_test_transcript_lengths_seg16s.csv", type=str)
    p.add_argument("--num-nodes", default=4, type=int)
    p.add_argument("--gpus", default=8, type=int)
    p.add_argument("--pretrained-model-path", default=None, type=str)
    p.add_argument("--transfer-frontend", action="store_true")
    p.add_argument("--transfer-encoder", action="store_true")
    p.add_argument("--warmup-epochs", default=5, type=int)
    p.add_argument("--max-epochs", default=75, type=int)
    p.add_argument("--max-frames", default=1600, type=int)
    p.add_argument("--lr", default=1e-3, type=float)
    p.add_argument("--weight-decay", default=0.03, type=float)
    p.add_argument("--ctc-weight", default=0.1, type=float)
    p.add_argument("--train-num-buckets", default=400, type=int)
    p.add_argument("--ckpt-path", default=None, type=str)
    p.add_argument("--slurm-job-id", default=0, type=float)
    p.add_argument("--debug", action="store_true")
    # native-driver extras (not in the reference)
    p.add_argument("--synthetic", action="store_true", help="train on the synthetic LRS3-shaped workload")
    p.add_argument("--steps", default=None, type=int, help="stop after this many optimizer steps")
    p.add_argument("--val-batches", default=8, type=int, help="synthetic validation batches per rank and epoch")
    p.add_argument("--synthetic-utterances", default=0, type=int, help="size of the synthetic corpus (default 20000)")
    p.add_argument("--numerics", default="mixed", choices=["mixed", "hpf", "precise", "bf16"],
                   help="numerical mode of the hot path (auto_avsr_amd.functional.set_mode); default: the mode bench.py times, "
                        "the cheapest one whose logits stay within 1e-3 of the fp32 reference")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of one replayed hipGraph per batch shape")
    p.add_argument("--trainer-step", default="auto", choices=["auto", "native"],
                   help="under a pytorch_lightning Trainer: auto = Lightning's automatic optimisation as in the reference (eager "
                        "launches, torch AdamW); native = manual optimisation, training_step replays the whole fused step as one "
                        "hipGraph per batch shape (auto_avsr_amd.train_native.NativeStepper: what bench.py times)")
    p.add_argument("--deterministic", action="store_true",
                   help="bit-reproducible training steps (the reference seeds everything, train.py:18): every gradient sum the default "
                        "build forms with floating-point atomics is formed in a fixed order instead; slower")
    p.add_argument("--grad-wire", default=None, choices=["f32", "bf16"],
                   help="format of the gradient buckets on the xGMI links (AVSR_DDP=buckets): f32 = the reference's DDP all-reduce "
                        "(default), bf16 = half the bytes per link, bf16 sums across the ranks")
    p.add_argument("--log-every", default=10, type=int)
    p.add_argument("--time-last", default=0, type=int, help="report the wall clock per step of the last N of --steps steps")
    return p.parse_args(argv)


def get_trainer(args):
    from pytorch_lightning import Trainer, seed_everything
    from pytorch_lightning.callbacks import LearningRateMonitor, ModelCheckpoint
    from pytorch_lightning.strategies import DDPStrategy

    seed_everything(42, workers=True)
    ckpt = ModelCheckpoint(dirpath=os.path.join(args.exp_dir, args.exp_name) if args.exp_dir else None,
                           monitor="monitoring_step", mode="max", save_last=True, filename="{epoch}", save_top_k=10)
    return Trainer(sync_batchnorm=True, default_root_dir=args.exp_dir, max_epochs=args.max_epochs,
                   num_nodes=args.num_nodes, devices=args.gpus, accelerator="gpu",
                   strategy=DDPStrategy(find_unused_parameters=False),
                   callbacks=[ckpt, LearningRateMonitor(logging_interval="step")],
                   reload_dataloaders_every_n_epochs=1,
                   # (manual optimisation: Lightning refuses gradient_clip_val; the fused step clips at 10 itself)
                   gradient_clip_val=None if getattr(args, "trainer_step", "auto") == "native" else 10.0)


def cli_main(argv=None):
    args = parse_args(argv)
    args.slurm_job_id = os.environ.get("SLURM_JOB_ID", args.slurm_job_id)  # optional here (reference: required)
    from lightning import HAVE_LIGHTNING, ModelModule

    if HAVE_LIGHTNING and not args.synthetic:
        from average_checkpoints import ensemble
        from datamodule.data_module import DataModule  # file-backed AVDataset: needs torchvision / torchaudio at read time

        module = ModelModule(args)
        trainer = get_trainer(args)
        trainer.fit(model=module, datamodule=DataModule(args, train_num_buckets=args.train_num_buckets),
                    ckpt_path=args.ckpt_path)
        ensemble(args)
        return
    from auto_avsr_amd.train_native import run

    run(args)


if __name__ == "__main__":
    cli_main()
