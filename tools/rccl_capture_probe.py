"""Which RCCL operations survive hipGraph capture on this stack (torch 2.10 + ROCm 7.2 RCCL), single-rank group on one GPU?
Every case runs in its own process under a set of environment switches; prints one line per case.
    python tools/rccl_capture_probe.py"""
import json
import os
import subprocess
import sys

CASES = ["allreduce", "allgather", "allreduce_sidestream", "ddp_backward"]
ENVS = [{}, {"TORCH_NCCL_ASYNC_ERROR_HANDLING": "0", "TORCH_NCCL_ENABLE_MONITORING": "0"},
        {"TORCH_NCCL_ASYNC_ERROR_HANDLING": "0", "TORCH_NCCL_ENABLE_MONITORING": "0", "TORCH_NCCL_AVOID_RECORD_STREAMS": "1",
         "TORCH_NCCL_DUMP_ON_TIMEOUT": "0", "TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC": "0"}]


def one(case):
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.ones(1 << 20, device=dev)
    y = torch.empty(1 << 20, device=dev)
    dist.all_reduce(x)  # communicator set-up outside capture
    dist.all_gather_into_tensor(y, x)
    torch.cuda.synchronize()
    mode = os.environ.get("PROBE_CAPTURE_MODE", "thread_local")
    g = torch.cuda.CUDAGraph()
    if case == "allreduce":
        with torch.cuda.graph(g, capture_error_mode=mode):
            x.mul_(2.0)
            dist.all_reduce(x)
            x.add_(1.0)
    elif case == "allgather":
        with torch.cuda.graph(g, capture_error_mode=mode):
            x.mul_(2.0)
            dist.all_gather_into_tensor(y, x)
            y.add_(1.0)
    elif case == "allreduce_sidestream":
        with torch.cuda.graph(g, capture_error_mode=mode):
            x.mul_(2.0)
            w = dist.all_reduce(x, async_op=True)
            w.wait()
            x.add_(1.0)
    else:
        lin = torch.nn.Linear(256, 256).to(dev)
        ddp = torch.nn.parallel.DistributedDataParallel(lin, device_ids=[0], gradient_as_bucket_view=True)
        inp = torch.randn(64, 256, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                ddp(inp).sum().backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode=mode):
            ddp(inp).sum().backward()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("CAPTURE_OK", float(x[0]))
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 2:
        one(sys.argv[1])
        sys.exit(0)
    for mode in ("thread_local", "global", "relaxed"):
        for ei, env in enumerate(ENVS):
            for case in CASES:
                e = dict(os.environ, PROBE_CAPTURE_MODE=mode, **env)
                try:
                    r = subprocess.run([sys.executable, __file__, case], capture_output=True, text=True, timeout=180, env=e)
                    ok = "CAPTURE_OK" in r.stdout
                    msg = "" if ok else (r.stderr.strip().splitlines() or [""])[-1][:160]
                    print(json.dumps({"mode": mode, "env": ei, "case": case, "ok": ok, "msg": msg}), flush=True)
                except subprocess.TimeoutExpired:
                    print(json.dumps({"mode": mode, "env": ei, "case": case, "ok": False, "msg": "timeout"}), flush=True)
