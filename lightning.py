"""ModelModule: the training / evaluation harness boundary of the reference (lightning.py:17-158) around the
MI355X-native ``E2E``.  With pytorch_lightning installed this is a LightningModule with the reference's hooks;
without it (this image) the same class is a plain nn.Module driven by ``auto_avsr_amd.train_native``."""
import torch

from cosine import WarmupCosineScheduler
from datamodule.transforms import TextTransform
from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E

try:  # optional third-party harness (absent in the build image)
    from pytorch_lightning import LightningModule as _Base

    HAVE_LIGHTNING = True
except ImportError:  # pragma: no cover - exercised in this image
    _Base = torch.nn.Module
    HAVE_LIGHTNING = False


def compute_word_level_distance(seq1, seq2):
    """Word-level Levenshtein distance (the reference calls torchaudio.functional.edit_distance, lightning.py:12-14)."""
    a, b = seq1.lower().split(), seq2.lower().split()
    prev = list(range(len(b) + 1))
    for i, wa in enumerate(a, 1):
        cur = [i]
        for j, wb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (wa != wb)))
        prev = cur
    return prev[-1]


def load_pretrained(model, args):
    """Weight-transfer modes of lightning.py:30-46 (front-end only / front-end + proj + encoder / full model)."""
    path = getattr(args, "pretrained_model_path", None)
    if not path:
        return
    ckpt = torch.load(path, map_location="cpu")
    if getattr(args, "transfer_frontend", False):
        sub = {k: v for k, v in ckpt["model_state_dict"].items() if k.startswith(("trunk.", "frontend3D."))}
        model.frontend.load_state_dict(sub)
    elif getattr(args, "transfer_encoder", False):
        for part in ("frontend", "proj_encoder", "encoder"):
            sub = {k[len(part) + 1:]: v for k, v in ckpt.items() if k.startswith(part + ".")}
            getattr(model, part).load_state_dict(sub)
    else:
        model.load_state_dict(ckpt)


class ModelModule(_Base):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if HAVE_LIGHTNING:
            self.save_hyperparameters(args)
        self.modality = args.modality
        self.text_transform = TextTransform()
        self.token_list = self.text_transform.token_list
        self.model = E2E(len(self.token_list), self.modality, ctc_weight=getattr(args, "ctc_weight", 0.1))
        load_pretrained(self.model, args)

    # ---- optimisation (lightning.py:48-52): AdamW(betas .9/.98) + per-step warm-up cosine
    def make_optimizer(self, steps_per_epoch):
        opt = torch.optim.AdamW(self.model.parameters(), lr=self.args.lr, weight_decay=self.args.weight_decay,
                                betas=(0.9, 0.98))
        sched = WarmupCosineScheduler(opt, self.args.warmup_epochs, self.args.max_epochs, steps_per_epoch)
        return opt, sched

    def configure_optimizers(self):
        n = len(self.trainer.datamodule.train_dataloader()) / self.trainer.num_devices / self.trainer.num_nodes
        opt, sched = self.make_optimizer(n)
        return [opt], [{"scheduler": sched, "interval": "step"}]

    # ---- the hot path (lightning.py:86-114)
    def _step(self, batch, batch_idx, step_type):
        loss, loss_ctc, loss_att, acc = self.model(batch["inputs"], batch["input_lengths"], batch["targets"])
        if HAVE_LIGHTNING:
            bs = len(batch["inputs"])
            sfx = "" if step_type == "train" else "_val"
            self.log("loss" + sfx, loss, on_step=step_type == "train", on_epoch=True, batch_size=bs,
                     sync_dist=step_type != "train")
            self.log("loss_ctc" + sfx, loss_ctc, on_step=False, on_epoch=True, batch_size=bs, sync_dist=True)
            self.log("loss_att" + sfx, loss_att, on_step=False, on_epoch=True, batch_size=bs, sync_dist=True)
            self.log("decoder_acc" + sfx, acc, on_step=step_type == "train", on_epoch=True, batch_size=bs, sync_dist=True)
            if step_type == "train":
                self.log("monitoring_step", torch.tensor(self.global_step, dtype=torch.float32))
        return loss

    def training_step(self, batch, batch_idx):
        loss = self._step(batch, batch_idx, "train")
        if HAVE_LIGHTNING:
            sizes = self.all_gather(batch["inputs"].size(0))
            loss = loss * (sizes.size(0) / sizes.sum())  # world size / total batch size (lightning.py:88-90)
        return loss

    def validation_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "val")

    # ---- evaluation (lightning.py:54-84,116-123): beam search over the encoder output
    def forward(self, sample):
        raise NotImplementedError("beam-search decoding (espnet.nets.batch_beam_search) is the section-8(f) 'next' "
                                  "item; the training hot path is E2E.forward")

    test_step = forward


def get_beam_search_decoder(model, token_list, rnnlm=None, rnnlm_conf=None, penalty=0, ctc_weight=0.1, lm_weight=0.0,
                            beam_size=40):
    raise NotImplementedError("beam search is outside this round's hot-path scope (SURVEY.md section 8f item 2)")
