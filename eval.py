"""Evaluation entry point with the reference's command line (eval.py:25-64): loads a checkpoint into ModelModule and
decodes with the hybrid CTC / attention beam search (auto_avsr_amd/decoding.py).  The reference iterates an LRS3 test
set through Lightning's Trainer.test; datasets and pytorch_lightning are not part of this image, so without them this
entry point decodes `--demo-frames` synthetic frames (plumbing check, BASELINE.json configs[0]) and prints the
hypothesis; with a DataModule available the WER loop is ModelModule.on_test_epoch_start / test_step / on_test_epoch_end."""
from argparse import ArgumentParser


def parse_args(argv=None):
    p = ArgumentParser()
    p.add_argument("--modality", type=str, default="video", choices=["audio", "video"])
    p.add_argument("--root-dir", type=str, default=None)
    p.add_argument("--test-file", default="lrs3_test_transcript_lengths_seg16s.csv", type=str)
    p.add_argument("--pretrained-model-path", type=str, default=None)
    p.add_argument("--decode-snr-target", type=float, default=999999)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--demo-frames", type=int, default=50, help="synthetic clip length when no dataset is given")
    return p.parse_args(argv)


def cli_main(argv=None):
    import torch

    from lightning import ModelModule

    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("eval.py needs an MI355X: the model runs on libavsr_hip.so only (no CPU path)")
    module = ModelModule(args).cuda().eval()
    if args.root_dir is not None:
        raise SystemExit("eval.py: dataset iteration needs the reference DataModule (torchaudio / torchvision / "
                         "pytorch_lightning), which this image does not have; ModelModule.test_step implements the WER loop")
    T = args.demo_frames
    sample = torch.randn(T, 1, 88, 88, device="cuda") if args.modality == "video" else torch.randn(T * 640, 1, device="cuda")
    with torch.no_grad():
        text = module(sample)
    print(f"hypothesis ({T} synthetic frames, random weights unless --pretrained-model-path): {text!r}")


if __name__ == "__main__":
    cli_main()
