"""CPU study (no GPU): where does the 16-bit forward lose its accuracy, and which operand format buys it back?

Runs the fp32 oracle (oracle/avsr_oracle.py) on the benchmarked batch A with the operands of chosen contraction groups
rounded to a 16-bit format before every product -- bf16 (8 significant bits, what the bf16 mode feeds the MFMA), f16 (11
bits) -- and compares decoder logits / CTC log-probabilities / encoder output with the REFERENCE numbers of
tests/golden/golden_bench_v1.pt exactly as tests/test_bench_parity.py does.  Accumulation stays f32 (as on the MFMA).

    python tools/precision_study.py                # the table of DESIGN.md section 2 (a few minutes on 8 cores)
    python tools/precision_study.py --config enc_ffn=f16,trunk=bf16
    python tools/precision_study.py --layers      # error GROWTH along the network: relative L2 of every block's output against the
                                                  # f32 oracle for bf16 / f16 / the default mixed policy (profiles/r4_layer_error_growth.txt)

Measurement script: imports oracle/ (allowed for tools, like tools/microbench_augment.py --cpu); never shipped."""
import argparse
import os
import sys
import time
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import avsr_oracle as O  # noqa: E402
from bench_common import BATCHES, FIXTURE, ODIM, bench_batch, bench_state_dict, full_errors, load_full, rel  # noqa: E402

GROUPS = ["stem", "trunk", "trunk1", "trunk2", "trunk3", "trunk4", "proj", "enc_ffn", "enc_ffn_w1", "enc_ffn_w2", "enc_attn_proj", "enc_attn_q", "enc_attn_k", "enc_attn_v",
          "enc_attn_out", "enc_attn_pos", "enc_attn_core", "enc_conv", "enc_conv_pw1", "enc_conv_dw", "enc_conv_pw2", "ctc_head", "dec", "dec_out"]
CFG = {}
STATS = {}


_exact_layer = [False]  # inside an encoder layer that `enc_exact_layers=N` keeps on exact (split-plane) arithmetic


def q(x, fmt, is_w=False):
    if _exact_layer[0]:
        return x
    if fmt == "f16a":
        fmt = None if is_w else "f16"
    if fmt == "f16w":
        fmt = "f16" if is_w else None
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt == "f16":
        return x.half().float()
    return x


def group_of(pre):
    if pre.startswith("proj_encoder"):
        return "proj"
    if pre.startswith("ctc."):
        return "ctc_head"
    if pre.startswith("decoder.output_layer"):
        return "dec_out"
    if pre.startswith("decoder."):
        return "dec"
    if ".feed_forward" in pre:
        fine = "enc_ffn_" + ("w1" if ".w_1." in pre else "w2")  # refine "enc_ffn" when named explicitly (round 6: per-input ablation)
        return fine if fine in CFG else "enc_ffn"
    if ".self_attn." in pre:
        fine = "enc_attn_" + pre.rstrip(".").rsplit("linear_", 1)[-1]  # q / k / v / out refine "enc_attn_proj" when named explicitly
        return fine if fine in CFG else "enc_attn_proj"
    if ".conv_module." in pre:
        return "enc_conv"
    return None


_scope = [None]


def note(group, x):
    m = float(x.abs().max())
    STATS[group] = max(STATS.get(group, 0.0), m)


def linear(sd, pre, x):
    g = group_of(pre)
    fmt = CFG.get(g)
    note(g, x)
    return F.linear(q(x, fmt), q(sd[pre + "weight"], fmt, True), sd.get(pre + "bias"))


class FShim(types.SimpleNamespace):
    """torch.nn.functional as seen by the oracle: convolutions and the position projection round their operands."""

    def __getattr__(self, name):
        return getattr(F, name)

    @staticmethod
    def _conv(fn, x, w, b=None, **kw):
        g = _scope[0]
        fmt = CFG.get(g)
        note(g, x)
        return fn(q(x, fmt), q(w, fmt, True), b, **kw)

    def conv1d(self, x, w, b=None, **kw):
        if _scope[0] == "enc_conv":  # refinements of the convolution module: pointwise 1 / depthwise / pointwise 2, when named
            sub = "enc_conv_dw" if kw.get("groups", 1) > 1 else ("enc_conv_pw1" if w.shape[0] == 2 * w.shape[1] else "enc_conv_pw2")
            if sub in CFG:
                note(sub, x)
                return F.conv1d(q(x, CFG[sub]), q(w, CFG[sub], True), b, **kw)
        return self._conv(F.conv1d, x, w, b, **kw)

    def conv2d(self, x, w, b=None, **kw):
        return self._conv(F.conv2d, x, w, b, **kw)

    def conv3d(self, x, w, b=None, **kw):
        return self._conv(F.conv3d, x, w, b, **kw)

    def linear(self, x, w, b=None):  # only linear_pos goes through F.linear directly
        fmt = CFG.get("enc_attn_pos", CFG.get("enc_attn_proj"))
        return F.linear(q(x, fmt), q(w, fmt), b)


class TorchShim:
    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def matmul(a, b):
        g = _scope[0]
        fmt = CFG.get(g)
        return torch.matmul(q(a, fmt), q(b, fmt))


def scoped(fn, group):
    def wrapper(*a, **kw):
        old = _scope[0]
        _scope[0] = group
        try:
            return fn(*a, **kw)
        finally:
            _scope[0] = old
    return wrapper


def install():
    O.linear = linear
    O.F = FShim()
    O.torch = TorchShim()
    O.conv_module = scoped(O.conv_module, "enc_conv")
    O.rel_mha = scoped(O.rel_mha, "enc_attn_core")
    O.mha = scoped(O.mha, "dec")
    orig_vf = O.video_frontend

    def video_frontend(sd, pre, x, train_bn=True):
        B, T = x.shape[0], x.shape[1]
        _scope[0] = "stem"
        y = x.transpose(1, 2)
        y = O.F.conv3d(y, sd[pre + "frontend3D.0.weight"], stride=(1, 2, 2), padding=(2, 3, 3))
        y = F.silu(O.batch_norm(sd, pre + "frontend3D.1.", y, train_bn))
        y = F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        y = y.transpose(1, 2).reshape(B * T, y.shape[1], y.shape[3], y.shape[4])
        for li in range(1, 5):
            _scope[0] = f"trunk{li}" if f"trunk{li}" in CFG else "trunk"
            for bi in range(2):
                stride = 2 if (li > 1 and bi == 0) else 1
                y = O.basic_block(sd, f"{pre}trunk.layer{li}.{bi}.", y, stride, train_bn)
        _scope[0] = None
        return y.mean(dim=(2, 3)).view(B, T, -1)

    O.video_frontend = video_frontend
    del orig_vf
    orig_layer = O.encoder_layer

    def encoder_layer(sd, pre, *a, **kw):
        n = int(pre.rstrip(".").rsplit(".", 1)[1])
        _exact_layer[0] = n < int(CFG.get("enc_exact_layers", 0)) or n >= 12 - int(CFG.get("enc_exact_last", 0))
        try:
            return orig_layer(sd, pre, *a, **kw)
        finally:
            _exact_layer[0] = False

    O.encoder_layer = encoder_layer


TRACE = None  # --layers: [(name, tensor)] outputs of the blocks, in execution order


def install_trace():
    def rec(fn, name_of):
        def wrapper(*a, **kw):
            y = fn(*a, **kw)
            if TRACE is not None:
                TRACE.append((name_of(a), y.detach().clone()))
            return y
        return wrapper

    O.basic_block = rec(O.basic_block, lambda a: "trunk " + a[1].split("trunk.")[1].rstrip("."))
    O.encoder_layer = rec(O.encoder_layer, lambda a: "encoder layer " + a[1].rstrip(".").rsplit(".", 1)[1])
    O.decoder_layer = rec(O.decoder_layer, lambda a: "decoder layer " + a[1].rstrip(".").rsplit(".", 1)[1])


def run(case, sd, batch, layer_probe=None):
    x, lengths, y = batch
    STATS.clear()
    with torch.no_grad():
        (loss, loss_ctc, loss_att, acc), mid = O.e2e_forward(sd, x, lengths, y)
        ctc = O.linear(sd, "ctc.ctc_lo.", mid["enc"])
    vcols, tsel = case["vcols"], case["tsel"]
    out = dict(loss=abs(float(loss) - case["loss"]) / abs(case["loss"]),
               ctc=abs(float(loss_ctc) - case["loss_ctc"]) / abs(case["loss_ctc"]),
               att=abs(float(loss_att) - case["loss_att"]) / abs(case["loss_att"]),
               dec_logits=rel(mid["pred"][:, :, vcols], case["dec_logits"]),
               ctc_logp=rel(torch.log_softmax(ctc, -1)[:, tsel][:, :, vcols], case["ctc_logp"]),
               enc=rel(mid["enc"][:, tsel, :32], case["enc"]), acc=acc)
    full = load_full(case["tag"])
    if full is not None:  # round 5: whole-tensor errors (decoder logits, RAW CTC logits, encoder output)
        fe = full_errors(full, mid["pred"], ctc, mid["enc"])
        out.update(dec_full=fe["dec_logits_full_rel_l2"], ctc_raw=fe["ctc_logits_raw_rel_l2"], enc_full=fe["enc_full_rel_l2"])
    return out, mid


def parse(spec):
    cfg = {}
    for item in spec.split(","):
        if not item:
            continue
        k, v = item.split("=")
        if k == "all":
            for g in GROUPS:
                if not g[-1].isdigit() and g not in ("enc_attn_q", "enc_attn_k", "enc_attn_v", "enc_attn_out", "enc_attn_pos", "enc_conv_dw", "enc_conv_pw1", "enc_conv_pw2"):  # refinements: only when named explicitly
                    cfg[g] = v
        elif k == "mixed":  # the round-4 default policy with format v
            for g in ("trunk3", "trunk4", "enc_ffn", "enc_attn_proj", "enc_attn_core", "enc_conv", "dec"):
                cfg[g] = v
        elif k == "enc":
            for g in ("enc_ffn", "enc_attn_proj", "enc_attn_core", "enc_conv"):
                cfg[g] = v
        elif k in ("enc_exact_layers", "enc_exact_last"):
            cfg[k] = v
        else:
            assert k in GROUPS, k
            cfg[k] = v
    return cfg


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", action="append", default=None)
    ap.add_argument("--batch", default="A")
    ap.add_argument("--layers", action="store_true")
    ap.add_argument("--ablate", action="store_true",
                    help="round 6 (VERDICT r5 item 4): the default mixed policy (f16 activations x exact weights) with ONE contraction "
                         "input of the encoder at a time kept exact in all 12 layers -- what a hi + lo split of that activation would buy "
                         "on the encoder output / raw CTC logits / decoder logits")
    args = ap.parse_args()
    if args.ablate:
        base = "mixed=f16a"
        args.config = [base] + [f"{base},{g}=f32" for g in ("enc_ffn_w1", "enc_ffn_w2", "enc_attn_q", "enc_attn_k", "enc_attn_v", "enc_attn_out",
                                                            "enc_attn_pos", "enc_attn_core", "enc_conv_pw1", "enc_conv_dw", "enc_conv_pw2",
                                                            "enc_ffn", "enc_attn_proj", "enc_conv", "dec", "trunk3", "trunk4")] + \
                      [f"{base},enc_ffn=f32,enc_attn_proj=f32,enc_attn_core=f32,enc_conv=f32"]
    torch.set_num_threads(os.cpu_count() or 8)
    if args.layers:
        install_trace()  # (innermost: the recorded outputs are those of the quantised blocks)
    install()
    fx = torch.load(FIXTURE, weights_only=False)
    case = fx[args.batch]
    cfgb = BATCHES[args.batch]
    sys.path.insert(0, ROOT)
    from tests.golden.synth import synth_state_dict  # noqa: F401,E402
    import json
    tmpl_path = os.path.join(ROOT, "tests", "golden", "golden_v1.pt")
    # the template state dict (names + shapes) comes from the product's module tree -- no GPU needed to build it
    from auto_avsr_amd.e2e import E2E
    m = E2E(ODIM, "video")
    sd = bench_state_dict(m.state_dict(), cfgb["seed"])
    batch = bench_batch(cfgb["lengths"], cfgb["L"], cfgb["seed"])
    assert cfgb.get("modality", "video") == "video", "the study wraps the video front-end only"
    if args.layers:
        MIXED = "trunk3=f16,trunk4=f16,enc_ffn=f16,enc_attn_proj=f16,enc_attn_core=f16,enc_conv=f16,dec=f16"  # functional.MIXED_POLICY
        cols = {}
        for name, spec in (("f32", ""), ("bf16", "all=bf16"), ("f16", "all=f16"), ("mixed", MIXED)):
            CFG.clear()
            CFG.update(parse(spec))
            TRACE = []
            out, mid = run(case, sd, batch)
            TRACE += [("encoder output (after_norm)", mid["enc"]), ("decoder logits", mid["pred"])]
            cols[name] = TRACE
            print(f"# {name}: logits vs REFERENCE golden {out['dec_logits']:.3e}", flush=True)
            TRACE = None
        print(f"# relative L2 error of every block's output against the f32 oracle, batch {args.batch} (tools/precision_study.py --layers)")
        print(f"{'block output':34s} {'bf16':>10s} {'f16':>10s} {'mixed':>10s}")
        for i, (nm, ref) in enumerate(cols["f32"]):
            e = [float((cols[k][i][1] - ref).norm() / ref.norm()) for k in ("bf16", "f16", "mixed")]
            print(f"{nm:34s} {e[0]:10.2e} {e[1]:10.2e} {e[2]:10.2e}")
        sys.exit(0)
    specs = args.config or ["", "all=bf16", "all=f16"] + [f"all=bf16,{g}=f32" for g in GROUPS] + \
        [f"{g}=bf16" for g in GROUPS] + [f"{g}=f16" for g in GROUPS]
    for spec in specs:
        CFG.clear()
        CFG.update(parse(spec))
        t0 = time.time()
        out, _ = run(case, sd, batch)
        print(json.dumps({"config": spec or "f32", **{k: (round(v, 8) if isinstance(v, float) else v) for k, v in out.items()},
                          "absmax": {k: round(v, 1) for k, v in STATS.items() if k}, "s": round(time.time() - t0, 1)}), flush=True)
