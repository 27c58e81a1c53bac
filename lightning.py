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


def steps_per_epoch(loader, world):
    """Optimizer steps per epoch and rank.  The reference divides the length of the UNSHARDED loader by the world size
    (lightning.py:49); this build's DataModule installs a DistributedSampler itself once a process group is up, and such a
    loader already reports the per-rank count -- dividing it again would end the cosine schedule after 1 / world of training."""
    from torch.utils.data.distributed import DistributedSampler

    if isinstance(getattr(loader, "sampler", None), DistributedSampler):
        return len(loader)
    return len(loader) / world


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
        # train.py --trainer-step native: Lightning's MANUAL optimisation -- training_step runs the whole step itself (forward,
        # backward, data-parallel exchange, fused clip + AdamW + schedule) through auto_avsr_amd.train_native.NativeStepper, i.e.
        # one replayed hipGraph per batch shape, the step bench.py times; the Trainer only feeds batches and runs its callbacks.
        # Default ("auto"): Lightning's automatic optimisation as in the reference (eager launches, torch AdamW, Trainer-side
        # clipping) -- measured 25.4 ms / step against 23.1 replayed in round 5.
        self.native_step = getattr(args, "trainer_step", "auto") == "native"
        self._native = None
        if self.native_step:
            self.automatic_optimization = False

    # ---- cross-rank BatchNorm (train.py:31 `sync_batchnorm=True`)
    def on_fit_start(self):
        """Lightning's `sync_batchnorm=True` swaps nn.BatchNorm modules for torch.nn.SyncBatchNorm, whose forward this
        build never calls (BatchNorm runs inside the fused HIP functions on the modules' parameters / buffers).  The
        cross-rank statistics are therefore switched on here, on the kernels' own path (functional.set_bn_sync: one
        all-gather per BatchNorm forward, one all-reduce per backward over RCCL)."""
        import torch.distributed as dist

        from auto_avsr_amd import functional as AF

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            AF.set_bn_sync(dist.group.WORLD)
        # the numerical mode of the hot path under a Lightning Trainer too: train.py's --numerics (default "mixed": what bench.py
        # times -- the cheapest arithmetic whose logits stay within 1e-3 of the fp32 reference); hipGraph replay per batch shape is
        # the native loop's (auto_avsr_amd.train_native): a Trainer drives backward / optimizer itself
        self._mode_before_fit = AF._save_mode()
        AF.set_mode(getattr(self.args, "numerics", None) or "mixed")
        if getattr(self, "native_step", False):
            from auto_avsr_amd.train_native import NativeStepper

            tr = self.trainer
            world = tr.num_devices * tr.num_nodes
            dev = next(self.model.parameters()).device
            n = steps_per_epoch(tr.datamodule.train_dataloader(), world)
            self._native = NativeStepper(self.model, self.args, dev, getattr(tr, "global_rank", 0), world, n)
            if getattr(self, "_native_opt_state", None) is not None:  # (a checkpoint loaded before the fit started)
                self._native.opt.load_state_dict(self._native_opt_state)
                self._native_opt_state = None

    def on_fit_end(self):
        from auto_avsr_amd import functional as AF

        if getattr(self, "_native", None) is not None:
            self._native.close()
            self._native = None
        AF.set_bn_sync(None)
        if getattr(self, "_mode_before_fit", None) is not None:
            AF._restore_mode(self._mode_before_fit)
            self._mode_before_fit = None

    # ---- optimisation (lightning.py:48-52): AdamW(betas .9/.98) + per-step warm-up cosine
    def make_optimizer(self, steps_per_epoch):
        opt = torch.optim.AdamW(self.model.parameters(), lr=self.args.lr, weight_decay=self.args.weight_decay,
                                betas=(0.9, 0.98))
        sched = WarmupCosineScheduler(opt, self.args.warmup_epochs, self.args.max_epochs, steps_per_epoch)
        return opt, sched

    def configure_optimizers(self):
        if self.native_step:
            return None  # manual optimisation: the fused optimizer lives inside the replayed step (on_fit_start); Lightning runs "with no optimizer"
        n = steps_per_epoch(self.trainer.datamodule.train_dataloader(), self.trainer.num_devices * self.trainer.num_nodes)
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

    def on_save_checkpoint(self, checkpoint):
        if self._native is not None:  # (Lightning saves no optimizer state under manual optimisation without optimizers)
            checkpoint["native_optimizer"] = self._native.opt.state_dict()

    def on_load_checkpoint(self, checkpoint):
        sd = checkpoint.get("native_optimizer")
        if sd is not None:
            if self._native is not None:
                self._native.opt.load_state_dict(sd)
            else:
                self._native_opt_state = sd

    def _native_training_step(self, batch):
        loss, loss_ctc, loss_att, hits, ntok = self._native(batch["inputs"], batch["input_lengths"], batch["targets"])
        if HAVE_LIGHTNING:
            bs = len(batch["inputs"])
            self.log("loss", loss, on_step=True, on_epoch=True, batch_size=bs)
            self.log("loss_ctc", loss_ctc, on_step=False, on_epoch=True, batch_size=bs, sync_dist=True)
            self.log("loss_att", loss_att, on_step=False, on_epoch=True, batch_size=bs, sync_dist=True)
            self.log("decoder_acc", hits / ntok.clamp_min(1), on_step=True, on_epoch=True, batch_size=bs, sync_dist=True)
            self.log("monitoring_step", torch.tensor(self.global_step, dtype=torch.float32))
        return loss

    def training_step(self, batch, batch_idx):
        from auto_avsr_amd import functional as AF

        if self._native is not None:
            return self._native_training_step(batch)
        AF.new_step()  # per-step registries of the kernels' autograd glue (zero-scratch arena, twin / hand-over tables)
        loss = self._step(batch, batch_idx, "train")
        if HAVE_LIGHTNING:
            sizes = self.all_gather(batch["inputs"].size(0))
            loss = loss * (sizes.size(0) / sizes.sum())  # world size / total batch size (lightning.py:88-90)
        return loss

    def validation_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "val")

    # ---- evaluation (lightning.py:54-84,116-123): beam search over the encoder output
    def _decode(self, sample):
        """Front-end -> encoder (no mask, B = 1) -> hybrid CTC/attention beam search -> text (lightning.py:54-64)."""
        x = self.model.frontend(sample.unsqueeze(0))
        x = self.model.proj_encoder(x)
        enc_feat, _ = self.model.encoder(x, None)
        enc_feat = enc_feat.squeeze(0)
        nbest_hyps = self.beam_search(enc_feat)
        nbest_hyps = [h.asdict() for h in nbest_hyps[: min(len(nbest_hyps), 1)]]
        predicted_token_id = torch.tensor(list(map(int, nbest_hyps[0]["yseq"][1:])))
        return self.text_transform.post_process(predicted_token_id).replace("<eos>", "")

    def decode_many(self, samples, workers=4):
        """Not in the reference: the transcripts of several utterances, their beam searches running concurrently (one host
        thread + stream + decoding session per worker, BatchBeamSearch.forward_many); front-end / encoder one utterance at a
        time as in `_decode`."""
        encs = []
        for sample in samples:
            x = self.model.proj_encoder(self.model.frontend(sample.unsqueeze(0)))
            enc_feat, _ = self.model.encoder(x, None)
            encs.append(enc_feat.squeeze(0))
        out = []
        for nbest in self.beam_search.forward_many(encs, workers=workers):
            ids = torch.tensor(list(map(int, nbest[0].asdict()["yseq"][1:])))
            out.append(self.text_transform.post_process(ids).replace("<eos>", ""))
        return out

    def forward(self, sample):
        self.beam_search = get_beam_search_decoder(self.model, self.token_list)
        return self._decode(sample)

    def on_test_epoch_start(self):
        self.total_length = 0
        self.total_edit_distance = 0
        self.text_transform = TextTransform()
        self.beam_search = get_beam_search_decoder(self.model, self.token_list)

    def test_step(self, sample, sample_idx):
        predicted = self._decode(sample["input"])
        actual = self.text_transform.post_process(sample["target"])
        self.total_edit_distance += compute_word_level_distance(actual, predicted)
        self.total_length += len(actual.split())

    def on_test_epoch_end(self):
        wer = self.total_edit_distance / max(self.total_length, 1)
        if HAVE_LIGHTNING:
            self.log("wer", wer)
        return wer


def get_beam_search_decoder(model, token_list, rnnlm=None, rnnlm_conf=None, penalty=0, ctc_weight=0.1, lm_weight=0.0,
                            beam_size=40):
    """lightning.py:126-158: decoder (1 - ctc_weight) + CTC prefix scorer (ctc_weight) + length bonus (penalty), beam 40,
    pre-beam on the decoder scores.  No language model is shipped with the reference (`scorers["lm"] = None`)."""
    from espnet.nets.batch_beam_search import BatchBeamSearch
    from espnet.nets.scorers.length_bonus import LengthBonus

    if rnnlm is not None or lm_weight != 0.0:
        raise NotImplementedError("language-model fusion is not part of the reference's released configuration")
    sos = eos = model.odim - 1
    scorers = model.scorers()
    scorers["lm"] = None
    scorers["length_bonus"] = LengthBonus(len(token_list))
    weights = {"decoder": 1.0 - ctc_weight, "ctc": ctc_weight, "lm": lm_weight, "length_bonus": penalty}
    return BatchBeamSearch(beam_size=beam_size, vocab_size=len(token_list), weights=weights, scorers=scorers, sos=sos, eos=eos,
                           token_list=token_list, pre_beam_score_key=None if ctc_weight == 1.0 else "decoder")
