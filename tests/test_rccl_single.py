"""The N > 1 path on RCCL with ONE rank (tools/rccl_world1.py): VERDICT r1 "the N>1 path has never executed on RCCL"."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_ddp_syncbn_fused_optimizer_on_single_rank_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py")], capture_output=True, text=True,
                       cwd=ROOT, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("\nRCCL world-1:", json.dumps(out))
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["grad_cos_min"] > 0.999
