"""GPU: run-to-run spread of the 50-step trajectory test's deviations (tests/test_trajectory.py) -- thresholds come from here."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

import test_trajectory as TT

ref = TT._fixture()["nodrop"]
for mode in ("mixed", "hpf"):
    for rep in range(4):
        rows, m = TT._run(torch.device("cuda"), mode, 50, False)
        dev = [abs(a["loss"] - b["loss"]) / max(abs(b["loss"]), 0.5) for a, b in zip(rows, ref["steps"])]
        sd = m.state_dict()
        cos = {k[-28:]: round(float(torch.dot(sd[k].detach().flatten()[:16].float().cpu(), v) / (sd[k].detach().flatten()[:16].float().cpu().norm() * v.norm())), 3)
               for k, v in ref["probe"].items() if "running_" not in k}
        print(mode, rep, "first5 max %.1e" % max(dev[:5]), "first10 max %.3f" % max(dev[:10]), "first20 mean %.3f" % (sum(dev[:20]) / 20),
              "all mean %.3f max %.2f" % (sum(dev) / 50, max(dev)), "tail loss %.3f acc %.2f" % (sum(r["loss"] for r in rows[-10:]) / 10, sum(r["acc"] for r in rows[-10:]) / 10),
              "gn5 %.3f" % max(abs(rows[s]["grad_norm"] - ref["steps"][s]["grad_norm"]) / ref["steps"][s]["grad_norm"] for s in range(5)), cos, flush=True)
