"""The reference's datamodule/transforms.py import path.
* TextTransform: token <-> text mapping (transforms.py:142-171).
* VideoTransform / AudioTransform / AdaptiveTimeMask / AddNoise: the reference runs these per sample in DataLoader
  workers on the CPU (torchvision / torchaudio); here the same names resolve to the device-side pipeline of
  auto_avsr_amd/transforms.py (csrc/augment.hip), which also offers the batch-level video_batch / audio_batch that fuse
  the transform with collate_pad."""
import os

import torch

from auto_avsr_amd.transforms import (AdaptiveTimeMask, AddNoise, AudioTransform, VideoTransform,  # noqa: F401
                                      audio_batch, pad_targets, video_batch)

_HERE = os.path.dirname(os.path.abspath(__file__))
SP_MODEL_PATH = os.path.join(os.path.dirname(_HERE), "spm", "unigram", "unigram5000.model")
DICT_PATH = os.path.join(os.path.dirname(_HERE), "spm", "unigram", "unigram5000_units.txt")


class TextTransform:
    """SentencePiece unigram-5000 tokenizer wrapper: ids 1..N from the units file, 0 = blank, N+1 = <eos>."""

    def __init__(self, sp_model_path=SP_MODEL_PATH, dict_path=DICT_PATH):
        self.spm = None
        if os.path.exists(sp_model_path):
            import sentencepiece

            self.spm = sentencepiece.SentencePieceProcessor(model_file=sp_model_path)
        if os.path.exists(dict_path):
            units = [line.split()[0] for line in open(dict_path, encoding="utf8").read().splitlines()]
        else:  # assets are data, not code: fall back to an anonymous vocabulary of the reference's size
            units = [f"<unit{i}>" for i in range(5047)]
        self.hashmap = {u: i + 1 for i, u in enumerate(units)}
        self.token_list = ["<blank>"] + units + ["<eos>"]
        self.ignore_id = -1

    def tokenize(self, text):
        tokens = self.spm.EncodeAsPieces(text)
        return torch.tensor([self.hashmap.get(t, self.hashmap["<unk>"]) for t in tokens])

    def post_process(self, token_ids):
        token_ids = token_ids[token_ids != -1]
        text = "".join(self.token_list[int(i)] for i in token_ids)
        return text.replace("▁", " ").strip().replace("<eos>", "")
