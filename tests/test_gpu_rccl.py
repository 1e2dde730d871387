"""RCCL transport smoke on the GPU box (world size 1 is all a 1-GPU box offers): the process group bench.py creates
(`backend="nccl"`, device_id) initialises, and the collectives of the source broadcast, the timing reduction and the barrier
run on a device tensor and return the right bytes.  Runs in a subprocess so that the process group never leaks into the
other tests' process."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import importlib.util, os, sys, torch, torch.distributed as dist
ROOT = sys.argv[1]
def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
D, WL = load("hg_dist", "dist.py"), load("hg_workloads", "workloads.py")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for (h, w) in ((37, 53), (2160, 3840)):
    want = torch.from_numpy(WL.lcg_image(w, h, 5)).to(dev)
    got = D.scatter_allgather(want.clone(), 0, 1, dist, verify=True)
    assert torch.equal(got, want)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.25
dist.destroy_process_group()
print("RCCL_OK")
"""


@pytest.mark.gpu
def test_rccl_world1_collectives():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.gpu
def test_bench_under_torchrun_world1():
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run, env rendezvous), with one rank."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "4", "--no-cpu-baseline", "--sources", "shared"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["verified"] is True
