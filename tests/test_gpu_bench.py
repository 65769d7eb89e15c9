"""The benchmark line of `python bench.py` at N=1: the fields the measurement contract names, the roofline block of
the dominant kernel, and the short runs of the other configurations that ride along."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*args, timeout=400):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _check_roofline(roof, with_digest):
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["kernel"].split("<")[0] in ("k_extend", "k_trace")
    stack = roof["traversal_stack"]
    assert 16 <= stack["entries_per_lane"] <= 64 and stack["need"] <= 64
    assert stack["guarded"] == (stack["entries_per_lane"] < stack["need"])  # fewer entries than the bound only if guarded
    assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    hbm, l1 = roof["hbm"], roof["l1_gather"]
    assert hbm["algorithmic_gbs"] == roof["achieved"] and hbm["algorithmic_frac"] == roof["frac"]
    assert roof["launches"] > 0 and roof["avg_launch_ms"] > 0 and roof["segments_per_launch"] > 0
    if with_digest:
        # HBM bytes per launch from the committed PMC digest of this workload, far below the algorithmic bytes:
        # the scene is served from the caches
        assert roof["traffic"] is not None and hbm["source"].startswith("profiles/")
        assert 0.0 < hbm["measured_frac"] < 1.0 and hbm["measured_gbs"] < hbm["algorithmic_gbs"]
        assert abs(roof["traffic"] - hbm["measured_bytes_per_segment"] * roof["segments_per_launch"]) < 1e-3 * roof["traffic"]
        assert roof["cache_resident"] is True
    assert l1["peak_gbs"] > 0 and abs(l1["frac"] - l1["achieved_gbs"] / l1["peak_gbs"]) < 1e-12
    assert l1["loads_per_segment"] > 10 and "r02_gather_peak.json" in l1["peak_source"]


def test_default_line_carries_the_contract_fields_and_the_secondary_runs():
    out = _bench("--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert out["unit"] == "Msamples/s" and out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f32" and "1920x1080" in out["config"]["workload"] and "cornell_box.gltf" in out["config"]["workload"]
    assert out["value"] > 2000.0  # the north-star floor, Msamples/s
    assert abs(out["value"] - out["traced_segments"] / (out["ms_per_step"] * 1e-3 * out["steps"]) / 1e6) < 1e-6 * out["value"]
    assert out["camera_paths"] == 1920 * 1080 * 64 * 2
    chk = out["image_check"]
    assert chk["alpha_equals_samples"] and chk["finite"] and chk["samples_per_pixel"] == 64 * 3 and chk["non_finite_pixels"] == 0
    assert out["launch_schedule_stable"] is True
    assert out["ranks_seen"] == 1 and out["distinct_gpus"] == 1 and out["ranks"][0]["device"]["pci"]
    _check_roofline(out["roofline"], with_digest=True)
    assert out["roofline"]["timed_bounces"] == [1, 8]
    sec = out["secondary"]
    assert [s["config"][:2] for s in sec] == ["C3", "C4", "C5"]
    for s in sec:
        assert s["value"] > 1000.0 and s["ms_per_step"] > 0 and s["unit"] == "Msamples/s"
        assert s["image_check"]["alpha_equals_samples"] and s["image_check"]["finite"]
        assert s["image_check"]["non_finite_pixels"] <= 8  # the reference's roulette: about one NaN sample per 1e8 paths
        _check_roofline(s["roofline"], with_digest=s["config"].startswith("C3"))
    assert "outside_synth" in sec[0]["workload"] and "3840x2160" in sec[1]["workload"] and "16 bounces" in sec[2]["workload"]


def test_cpu_baseline_states_the_cores_it_used():
    out = _bench("--steps", "1", "--warmup", "1", "--no-secondary", "--cpu-frames", "1", "--width", "480", "--height", "270")
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "Msamples/s" and cpu["value"] > 0
    assert cpu["cores"] >= 1 and cpu["nproc"] == os.cpu_count() and cpu["cores"] <= cpu["nproc"]
    assert cpu["cores"] in (cpu["cores_this_process_may_use"], int(os.environ.get("OMP_NUM_THREADS", "0") or 0))
    assert "nproc" in cpu["sample"]
