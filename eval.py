"""Evaluation entry point with the reference's command line (eval.py:25-79): loads a checkpoint into ModelModule and
runs the WER loop -- hybrid CTC / attention beam search per utterance (auto_avsr_amd/decoding.py), word-level edit
distance against the reference transcript, WER = total distance / total words (lightning.py:69-84,116-123).

With pytorch_lightning installed the loop is driven by `Trainer.test(model, datamodule)` exactly as in the reference.
Without it (this image) the same three hooks -- on_test_epoch_start / test_step / on_test_epoch_end -- are called
directly over `DataModule.test_dataloader()`.  `--root-dir` selects the reference's on-disk test set; without it the
loader yields `--synthetic-utterances` LRS3-shaped synthetic utterances (plumbing check: random targets, so the WER of an
untrained model is ~1; BASELINE.json configs[0])."""
import logging
from argparse import ArgumentParser


def parse_args(argv=None):
    p = ArgumentParser()
    p.add_argument("--modality", type=str, default="video", choices=["audio", "video"])
    p.add_argument("--root-dir", type=str, default=None)
    p.add_argument("--test-file", default="lrs3_test_transcript_lengths_seg16s.csv", type=str)
    p.add_argument("--pretrained-model-path", type=str, default=None)
    p.add_argument("--decode-snr-target", type=float, default=999999)
    p.add_argument("--debug", action="store_true")
    # extras of this build (not in the reference)
    p.add_argument("--synthetic-utterances", type=int, default=0,
                   help="decode this many synthetic utterances instead of a dataset (default when --root-dir is absent: 4)")
    p.add_argument("--max-test-frames", type=int, default=100, help="length cap of the synthetic utterances")
    p.add_argument("--numerics", choices=["precise", "mixed", "bf16"], default="precise",
                   help="arithmetic of the decoding forward pass: precise (split hi / lo bf16 planes -- hypotheses equal an fp32 run "
                        "of the reference; the default) or bf16 (faster on long utterances)")
    p.add_argument("--decode-workers", type=int, default=1,
                   help="beam searches in flight at once (host threads + streams, one decoding session each); 1 = the reference's "
                        "one-utterance-at-a-time loop")
    return p.parse_args(argv)


def run_test_loop(module, loader, device, log=None, decode_workers=1):
    """Trainer.test without Lightning: the module's own hooks over the loader (lightning.py:69-84,116-123).  decode_workers > 1:
    utterances are taken in groups whose beam searches run concurrently (ModelModule.decode_many); same transcripts, same WER."""
    import torch

    from lightning import compute_word_level_distance

    module.on_test_epoch_start()
    with torch.no_grad():
        if decode_workers <= 1:
            for i, sample in enumerate(loader):
                sample = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sample.items()}
                module.test_step(sample, i)
                if log is not None:
                    log(i, module.total_edit_distance, module.total_length)
        else:
            group, done = [], 0

            def flush():
                nonlocal done
                for s, predicted in zip(group, module.decode_many([s["input"] for s in group], workers=decode_workers)):
                    actual = module.text_transform.post_process(s["target"])
                    module.total_edit_distance += compute_word_level_distance(actual, predicted)
                    module.total_length += len(actual.split())
                    if log is not None:
                        log(done, module.total_edit_distance, module.total_length)
                    done += 1
                group.clear()

            for sample in loader:
                group.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sample.items()})
                if len(group) == 4 * decode_workers:
                    flush()
            if group:
                flush()
    return module.on_test_epoch_end()


def cli_main(argv=None):
    import torch

    from datamodule.data_module import DataModule
    from lightning import HAVE_LIGHTNING, ModelModule

    args = parse_args(argv)
    from auto_avsr_amd import functional as AF

    AF.set_mode(args.numerics)  # evaluation is forward-only: "precise" here is also what the hpf training mode's forward runs
    logging.basicConfig(format="%(asctime)s %(message)s" if args.debug else "%(message)s",
                        level=logging.DEBUG if args.debug else logging.INFO, datefmt="%Y-%m-%d %H:%M:%S")
    if not torch.cuda.is_available():
        raise SystemExit("eval.py needs an MI355X: the model runs on libavsr_hip.so only (no CPU path)")
    if args.root_dir is None and not args.synthetic_utterances:
        args.synthetic_utterances = 4
    module = ModelModule(args)
    datamodule = DataModule(args)
    if HAVE_LIGHTNING and not args.synthetic_utterances:
        from pytorch_lightning import Trainer

        Trainer(num_nodes=1, devices=1, accelerator="gpu").test(model=module, datamodule=datamodule)
        return
    module = module.cuda().eval()
    if args.synthetic_utterances:
        from auto_avsr_amd.synthetic import utterance_lengths
        from datamodule.av_dataset import SyntheticAVDataset

        lens = [min(int(t), args.max_test_frames) for t in utterance_lengths(args.synthetic_utterances, seed=7)]
        loader = torch.utils.data.DataLoader(
            SyntheticAVDataset(len(lens), args.modality, odim=module.model.odim, seed=2, lengths=lens), batch_size=None)
    else:
        loader = datamodule.test_dataloader()
    wer = run_test_loop(module, loader, torch.device("cuda"), decode_workers=args.decode_workers,
                        log=lambda i, d, n: logging.info(f"utt {i}: running WER {d / max(n, 1):.4f} ({d}/{n} words)"))
    print(f"WER {wer:.4f} over {module.total_length} reference words")
    return wer


if __name__ == "__main__":
    cli_main()
