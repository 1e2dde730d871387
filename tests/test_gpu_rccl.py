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


def _torchrun_bench(extra, timeout=900):
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_strong_scaling_under_torchrun_world1():
    """north_star's wording -- a FIXED batch split over the ranks -- through the code path the multi-GPU lines take: `--scaling strong`
    under torch.distributed.run (one rank: all a 1-GPU box offers).  The batch is what the ranks warped, the timed bytes verify against
    the reference's goldens, end_to_end prices the fan-out, and the roofline block carries the <= 1 fabric figure's fields."""
    out = _torchrun_bench(["--scaling", "strong", "--batch", "8", "--config", "C4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sources", "shared"])
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["verified"] is True
    cfg = out["config"]
    assert cfg["frames_per_step_all_gpus"] == 8 and cfg["frames_per_gpu_per_step"] is None
    assert [r["frames"] for r in cfg["ranks"]] == [8] and cfg["ranks"][0]["first_frame"] == 0
    assert cfg["end_to_end"]["ms_per_batch_incl_broadcast"] >= out["ms_per_step"]
    assert "STRONG" in cfg["value_is"]
    rf = out["roofline"]
    assert rf["write_floor_ms"] > 0 and "fabric_frac" in rf and rf["kernel_ms"] > 0


@pytest.mark.gpu
def test_bench_also_entries_under_torchrun_world1():
    """`--also CONFIG:FRAMES` and `CONFIG:BATCH:strong` entries (what the plain command adds by default): every entry is a complete,
    verified line of its own with the scaling it names."""
    out = _torchrun_bench(["--steps", "3", "--warmup", "1", "--frames", "4", "--no-cpu-baseline", "--sources", "shared", "--points", "resident",
                           "--also", "C2:4,C5:2:strong"])
    assert out["verified"] is True and out["scaling"] == "weak" and "WEAK" in out["config"]["value_is"]
    also = out["also"]
    assert [a["metric"] for a in also] == ["Mpixels/s warped (C2)", "Mpixels/s warped (C5)"]
    assert also[0]["scaling"] == "weak" and also[0]["config"]["frames_per_gpu_per_step"] == 4 and also[0]["verified"] is True
    assert also[1]["scaling"] == "strong" and also[1]["config"]["frames_per_step_all_gpus"] == 2 and also[1]["verified"] is True
    assert also[1]["roofline_distinct"] is None and also[1]["roofline_fresh"] is None      # strong entries: shared source, resident points
