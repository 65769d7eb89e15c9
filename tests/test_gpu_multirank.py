"""The multi-rank path on the one GPU of the test box: the RCCL ("nccl") gather executed for real in a
one-rank process group, and bench.py launching its own ranks (two ranks sharing the GPU, gloo rendezvous)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_RCCL_WORLD1 = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
import logipathtracer_amd as L
from logipathtracer_amd.gather import gather_image
from logipathtracer_amd.partition import Partition

torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % int(sys.argv[1]), rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
scene = L.Scene(os.path.join("tests", "golden", "cornell_box.gltf"))
w, h = 200, 120
ubos = L.frame_ubos(scene.camera_world, scene.fov_y, 2, seed=5)
dev = L.Device(0)
dev.upload_scene(scene)
dev.resize(w, h)
dev.set_partition(32, 32, 0, 1)      # tiles that stick out of the image: the compact buffer has unused slots
dev.set_options(4, L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
dev.render_frames(ubos)
want = dev.read_accum()
image, ms, nbytes = gather_image(dev, 1, w, h, torch.device("cuda", 0), "nccl")
assert nbytes == dev.owned_slots() * 16
got = image.cpu().numpy()
assert got.shape == want.shape and np.array_equal(got, want), "gathered image differs"
assert np.all(got[..., 3] == 2.0)
# a second collective on the same group: a reduction as bench.py does for its timings
t = torch.tensor([1.5, float(dev.stats().segments)], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t[0]) == 1.5
# the per-rank reports travel through all_gather_object on the same group
from logipathtracer_amd import ranks as R
st = dev.stats()
reports = R.collect(R.rank_report(0, 0, 0, {"device_ms": st.device_ms, "segments": st.segments, "paths": st.paths, "elapsed": 0.1},
                                  dev.owned_slots()))
assert len(reports) == 1 and reports[0]["device"]["pci"] and reports[0]["device"]["uuid"] and R.shared_devices(reports) == []
dist.barrier()
dist.destroy_process_group()
dev.close()
print("rccl world-1 gather ok, %.3f ms" % ms)
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_rccl_gather_in_a_one_rank_group():
    """all_gather_into_tensor over RCCL on the torch alias of lpt_owned_accum_device, then
    lpt_assemble_gathered: the image equals lpt_read_accum's."""
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1, str(_free_port())], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl world-1 gather ok" in r.stdout


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks, shards the image between them,
    gathers it and prints one JSON line (ranks share the box's one GPU, so the gather goes through gloo)."""
    env = _env()
    env["LPT_BENCH_BACKEND"] = "gloo"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--spp-per-step", "2", "--width", "320", "--height", "184", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["camera_paths"] == 320 * 184 * 4 * 1  # 2 spp per GPU x 2 ranks, one timed step
    assert out["gather_ms"] is not None and out["config"]["tile"] == 64
    # evidence that two ranks took part, where, and how evenly
    assert out["ranks_seen"] == 2 and [r["rank"] for r in out["ranks"]] == [0, 1]
    for r in out["ranks"]:
        assert r["device"]["device_index"] == 0 and r["device_ms"] > 0 and r["segments"] > 0 and r["owned_slots"] > 0
    assert out["ranks"][0]["pid"] != out["ranks"][1]["pid"]
    assert out["distinct_gpus"] == 1 and out["backend"] == "gloo"  # the rehearsal shares the one GPU of the box
    assert out["imbalance"] >= 1.0
    assert out["gather_bytes"] == 2 * out["ranks"][0]["owned_slots"] * 16
    assert sum(r["segments"] for r in out["ranks"]) == out["traced_segments"]
    assert out["image_check"]["alpha_equals_samples"] and out["image_check"]["finite"]
    assert out["image_check"]["samples_per_pixel"] == 8.0  # 4 spp per step, one warm-up and one timed step


def test_bench_strong_scaling_keeps_the_total_samples():
    """--scaling strong: spp per step does not grow with the ranks, so each rank's batch shrinks."""
    env = _env()
    env["LPT_BENCH_BACKEND"] = "gloo"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--spp-per-step", "2", "--width", "320", "--height", "184", "--no-cpu-baseline", "--scaling", "strong"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["spp_per_step"] == 2
    assert out["camera_paths"] == 320 * 184 * 2


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 2" in r.stderr
