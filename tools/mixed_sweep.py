"""Forward-format policy sweep of the mixed numerical mode on the MI355X: parity of the full-size video model against the
reference goldens at the survey's batch A and batch B (tests/golden/golden_bench_v1.pt) for a list of policies
(functional.MIXED_POLICY: component -> "f16" | "f16x2" | "split"); round 5: whole-tensor errors first.  One model per batch, re-used by every policy.
    python tools/mixed_sweep.py [policy ...]          # policy = "encoder=f16,trunk1=f16,..." ("" = everything on split planes)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

import bench_common as BC
from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e import E2E

T4 = "trunk1=f16,trunk2=f16,trunk3=f16,trunk4=f16"
DEFAULT = ["encoder=f16", f"encoder=f16,{T4}", f"encoder=f16,{T4},decoder=f16", f"encoder=f16,{T4},decoder=f16,dec_out=f16",
           "encoder=f16,trunk2=f16,trunk3=f16,trunk4=f16,decoder=f16,dec_out=f16", "encoder=f16,decoder=f16,dec_out=f16",
           f"encoder=f16,{T4},dec_out=f16"]
tags = ("A", "B")
if len(sys.argv) > 1 and sys.argv[1].startswith("--tags="):  # e.g. --tags=AA (the audio model at batch A's geometry)
    tags = tuple(sys.argv.pop(1)[7:].split(","))
policies = sys.argv[1:] or DEFAULT
gold = torch.load(BC.FIXTURE, weights_only=False)
for tag in tags:
    case = gold[tag]
    AF.invalidate_weight_cache()
    m = E2E(BC.ODIM, case.get("modality", "video"))
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    sd = BC.bench_state_dict(m.state_dict(), case["seed"])
    m.load_state_dict(sd)
    m = m.cuda().train()
    for pol in policies:
        AF.MIXED_POLICY = dict(kv.split("=") for kv in pol.split(",") if kv)
        AF.invalidate_weight_cache()
        m.load_state_dict(sd)  # (BatchNorm running statistics back to the start)
        with AF.numerics("mixed"):
            r = BC.measure(m, case, torch.device("cuda"))
        print(json.dumps({"batch": tag, "policy": pol, **{k: float(f"{r[k]:.3g}") for k in (
            "dec_logits_full_rel_l2", "ctc_logits_raw_rel_l2", "enc_full_rel_l2", "dec_logits_rel_l2", "ctc_logp_rel_l2", "loss_rel_err",
            "grad_sample_cos_min", "grad_sample_rel_l2_median") if k in r}}), flush=True)
