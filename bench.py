#!/usr/bin/env python3
"""Benchmark of the RendererPT compute path on MI355X.

Metric (BASELINE.json): Msamples/s at 1920x1080, 8-bounce Cornell box, where a sample is one
ray segment actually traced (one iteration of traceRay's loop, shaders/path_tracing.comp:216-324).
`value` counts traced segments only; the nominal figure W*H*spp*bounces/t is reported beside it.

A step renders `spp_per_step` frames (one sample per pixel each) of the whole image.  With N GPUs the
image is cut into tiles dealt round-robin to the ranks.  `--scaling weak` (default): spp_per_step grows
with N, so each GPU traces the same number of camera paths per step at every N.  `--scaling strong`:
spp_per_step stays what it is at N=1, so each GPU's batch shrinks with N.  No collective runs in the
timed region; the single gather of the HDR tile buffers happens after it and is timed apart.

At N=1 the line also carries `secondary`: short runs of the other benchmark configurations that fit one
GPU (the open scene at 1080p, the 4K resolutions), each with its own roofline block, and `cpu_baseline`.

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E
# Nominal rate of the vector L1, the path every per-lane gather of the search takes: 64 B per clock and CU at
# the peak engine clock.  A derived figure, kept as a note only: the ceiling the line prices against is the
# measured one (profiles/r02_gather_peak.json).
L1_BYTES_PER_CLK_PER_CU = 64.0
PEAK_CLOCK_GHZ = 2.4

CORNELL = os.path.join(ROOT, "tests", "golden", "cornell_box.gltf")
PIPELINE_NAMES = ["wavefront", "megakernel", "persistent", "auto", "hybrid"]
BUILDER_NAMES = ["split_bvh", "binned_sah", "lbvh_gpu", "ploc_gpu"]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: starts N ranks of this script under
    torch.distributed.run (one per GPU) as a child process, relays their output and returns the exit
    status.  Runs before anything in this process touches a GPU."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--spp-per-step", type=int, default=64,
                    help="frames per step: per GPU with --scaling weak, in total with --scaling strong")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--scene", default=CORNELL)
    ap.add_argument("--builder", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=3, help="0 wavefront, 1 megakernel, 2 persistent, 3 automatic, 4 hybrid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other configurations (N=1)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample (0: automatic)")
    return ap.parse_args()


def host_threads():
    """Threads the CPU legs use: every core this process may run on (OMP_NUM_THREADS if the box sets it)."""
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    return (env or affinity), affinity


def oracle_scene(scene):
    from oracle import binding as O
    return O.OracleScene(scene.objects, scene.object_bvh, scene.vertices, scene.mesh_bvh, getattr(scene, "textures", ()))


def cpu_baseline(scene, ubos, width, height, bounces, frames):
    """Times the CPU oracle (OpenMP over pixels) on a bounded sample of the same workload and returns
    (baseline dict, oracle work counters of the sample)."""
    from oracle import binding as O
    osc = oracle_scene(scene)
    threads, affinity = host_threads()
    if frames <= 0:
        t0 = time.perf_counter()
        O.render_frames(osc, ubos[:1], width, height, bounces, threads=threads)
        one = time.perf_counter() - t0
        frames = int(max(1, min(len(ubos), 15.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    _, st = O.render_frames(osc, ubos[:frames], width, height, bounces, threads=threads)
    dt = time.perf_counter() - t0
    base = {"value": st["segments"] / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "nproc": os.cpu_count(), "cores_this_process_may_use": affinity,
            "sample": f"{frames} frame(s) of {width}x{height}, {bounces} bounces, same scene and seeds "
                      f"({st['segments']} segments in {dt:.2f} s); OpenMP over pixel rows on every core this process may "
                      f"use (nproc = {os.cpu_count()})"}
    return base, st


def traversal_counters(scene, ubos, width, height, bounces):
    """Oracle work counters (nodes and triangles per segment) from one frame at a quarter of the
    resolution: enough for the per-segment means the roofline needs, in well under a second."""
    from oracle import binding as O
    threads, _ = host_threads()
    _, st = O.render_frames(oracle_scene(scene), ubos[:1], max(16, width // 4), max(16, height // 4), bounces, threads=threads)
    return st


def range_means(upto_hi, upto_lo):
    """Per-segment means over the segments of bounces [lo, hi), from the counters of oracle runs cut after
    hi and after lo segments (both normalised per camera path, so the runs may differ in size; upto_lo is
    None for lo = 0)."""
    ph = max(upto_hi["paths"], 1)
    pl = max(upto_lo["paths"], 1) if upto_lo else 1
    lo = (lambda k: upto_lo[k] / pl) if upto_lo else (lambda k: 0.0)
    seg = upto_hi["segments"] / ph - lo("segments")  # segments of the range per camera path
    out = {k: (upto_hi[k] / ph - lo(k)) / max(seg, 1e-9) for k in ("boxTests", "rootVisits", "triTests", "hits", "objectsEntered")}
    out["segments_per_path"] = seg
    return out


def workload_key(scene_path, width, height, bounces):
    """Name under which the committed PMC digests of a workload are filed (profiles/r<NN>_traffic_<key>.json)."""
    base = os.path.basename(scene_path)
    scene = "cornell" if base == "cornell_box.gltf" else ("outside" if base.startswith("outside_synth") else None)
    if scene is None:
        return None
    return f"{scene}_{width}x{height}_{bounces}b"


def traffic_profile(key, kernel, pipeline):
    """The newest committed digest of rocprofv3's HBM counters for this workload and kernel, or None."""
    if key is None:
        return None, None
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{key}.json")):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None, None
    try:
        t = json.load(open(best[1]))
    except Exception:
        return None, None
    if t.get("kernel") != kernel or t.get("pipeline", pipeline) != pipeline:
        return None, None
    return t, "profiles/" + os.path.basename(best[1])


def roofline(wl, scene, ubos, counters, used_pipeline, kernel_seg, ext_ms, ext_launches, n_cus, timed_bounces, stack=None):
    """Roofline block of the dominant kernel: k_extend (wavefront organisation) or k_trace.

    Top level, as the measurement contract defines it: bound = HBM, achieved = algorithmic bytes per traced
    segment (48 B per interchange node fetched, both tested children included, 36 B per triangle tested, plus the
    kernel's ray/hit or surface state) x the segments the kernel traced / its time, the time measured live with HIP
    events on the context's stream; peak = 8 TB/s; traffic = HBM bytes per launch from the committed rocprofv3 PMC
    digest of the same workload (null without one: counters cannot be read from inside this process).
    The algorithmic rate exceeds the HBM peak wherever the scene is served from the caches, so two sub-blocks say
    what does hold: `hbm` (algorithmic and measured rates side by side) and `l1_gather` (the per-lane 16-byte loads
    the search issues against the measured ceiling of such gathers on this chip, tools/gather_peak.hip)."""
    kernel = {0: "k_extend", 2: "k_trace", 4: "k_trace"}.get(used_pipeline, "k_megakernel")
    # the bounces whose segments the timed kernel traces: [lo, hi) (lpt_stats; the wavefront organisation hands
    # the late bounces to one persistent launch where the queues become short)
    lo, hi = timed_bounces
    w, h, bounces = wl["width"], wl["height"], wl["bounces"]
    upto_hi = counters if hi == bounces else traversal_counters(scene, ubos, w, h, hi)
    upto_lo = traversal_counters(scene, ubos, w, h, lo) if lo > 0 else None
    rm = range_means(upto_hi, upto_lo)
    n_inner = rm["boxTests"] / 2.0  # an inner node's visit tests both of its children
    n_node = rm["boxTests"] + rm["rootVisits"]
    n_tri, p_hit, n_obj = rm["triTests"], rm["hits"], rm["objectsEntered"]
    kernel_seg = float(kernel_seg)
    # paths that enter the kernel, per camera path: every camera path at bounce 0, else the first hits that went on
    entering = 1.0 if lo == 0 else upto_lo["hits"] / max(upto_lo["paths"], 1)
    paths_per_seg = entering / max(rm["segments_per_path"], 1e-9)
    search_loads = 4.0 * n_inner + 3.0 * n_tri + 4.0 * n_obj
    if kernel == "k_extend":
        other_loads = 2.0                 # the queued ray: origin and direction
        state_bytes = 24.0 + 12.0
        state_note = "36 B ray/hit"
        loads_note = "2 for the queued ray"
    else:
        # per hit the surface's records: 8 loads of the object's shading record, 6 of the triangle's vertices
        # (positions and normals); per path taken off the queue 3 loads (hybrid organisation)
        per_path_loads = 3.0 if used_pipeline == 4 else 0.0
        other_loads = p_hit * 14.0 + per_path_loads * paths_per_seg
        per_path = 16.0 + (48.0 if used_pipeline == 4 else 0.0)
        state_bytes = p_hit * (208.0 + 144.0) + per_path * paths_per_seg
        state_note = f"352 B object+vertices per hit + {per_path:.0f} B per path"
        loads_note = "14 per hit for the surface's records" + (" + 3 per queued path" if per_path_loads else "")
    loads_per_seg = search_loads + other_loads
    secs = ext_ms * 1e-3
    launches = max(ext_launches, 1)
    bytes_per_seg = 48.0 * n_node + 36.0 * n_tri + state_bytes
    algorithmic_gbs = bytes_per_seg * kernel_seg / secs / 1e9
    gather_gbs = 16.0 * loads_per_seg * kernel_seg / secs / 1e9
    nominal_l1 = n_cus * L1_BYTES_PER_CLK_PER_CU * PEAK_CLOCK_GHZ

    hbm = {"peak_gbs": HBM_PEAK_GBS, "algorithmic_bytes_per_segment": bytes_per_seg, "algorithmic_gbs": algorithmic_gbs,
           "algorithmic_frac": algorithmic_gbs / HBM_PEAK_GBS,
           "algorithmic_note": f"48 B/node fetched + 36 B/triangle tested + {state_note}; a fraction above 1 means the scene is "
                               "served from L1/L2, not from HBM",
           "measured_bytes_per_segment": None, "measured_gbs": None, "measured_frac": None, "source": None}
    traffic = None
    prof, source = traffic_profile(workload_key(wl["scene_path"], w, h, bounces), kernel, used_pipeline)
    if prof:
        per_seg = prof["hbm_bytes_per_launch"] / prof["segments_per_launch"]
        measured_gbs = per_seg * kernel_seg / secs / 1e9
        traffic = per_seg * kernel_seg / launches
        hbm.update({"measured_bytes_per_segment": per_seg, "measured_gbs": measured_gbs,
                    "measured_frac": measured_gbs / HBM_PEAK_GBS,
                    "source": f"{source}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                              f"{prof.get('rule', '')}; bytes per segment from there, segments and time from this run",
                    "unit_busy_in_profile": prof.get("units")})
    l1 = {"achieved_gbs": gather_gbs, "loads_per_segment": loads_per_seg,
          "nominal_peak_gbs": nominal_l1, "nominal_frac": gather_gbs / nominal_l1,
          "nominal_note": f"{n_cus} CUs x 64 B/clk x {PEAK_CLOCK_GHZ:.1f} GHz: derived, not a figure of the hardware guide",
          "note": f"16 B x per-lane loads ({loads_per_seg:.1f} per segment: 4 per inner node, 3 per triangle, 4 per object "
                  f"entered, {loads_note}) x segments / kernel time"}
    gpath = os.path.join(ROOT, "profiles", "r02_gather_peak.json")
    if os.path.exists(gpath):
        try:
            g = json.load(open(gpath))["ceilings_gbs"]
            l1.update({"peak_gbs": g["records_64B_l1_resident_16KiB_table"], "frac": gather_gbs / g["records_64B_l1_resident_16KiB_table"],
                       "l2_resident_peak_gbs": g["records_64B_l2_resident_2MiB_table"],
                       "peak_source": "profiles/r02_gather_peak.json (tools/gather_peak.hip on this chip: random 64-byte records "
                                      "from an L1-resident table, every lane its own; lanes of the search that meet at one node "
                                      "share a line, so the kernel may exceed this random-address ceiling)"})
        except Exception:
            pass
    return {
        "kernel": kernel + (f"<{stack['entries_per_lane']}{', guarded' if stack.get('guarded') else ''}>" if stack else ""), "bound": "hbm", "achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": algorithmic_gbs / HBM_PEAK_GBS, "traffic": traffic,
        "cache_resident": bool(algorithmic_gbs > HBM_PEAK_GBS or (hbm["measured_gbs"] or algorithmic_gbs) < 0.5 * algorithmic_gbs),
        "note": f"achieved = algorithmic bytes of the segments {kernel} traced (bounces {lo} to {hi - 1}) / its time (HIP events "
                "around every launch on the context's stream); traffic = HBM bytes per launch by the PMC digest named in "
                "hbm.source.  The kernel is bound by the L1 gather path and vector issue, not by HBM: see l1_gather",
        "timed_bounces": [lo, hi], "traversal_stack": stack,
        "avg_launch_ms": ext_ms / launches, "launches": ext_launches, "segments_per_launch": kernel_seg / launches,
        "inner_nodes_per_segment": n_inner, "nodes_fetched_per_segment": n_node, "tris_per_segment": n_tri,
        "objects_entered_per_segment": n_obj, "hits_per_segment": p_hit,
        "hbm": hbm, "l1_gather": l1,
    }


def measure(dev, ubos, spp_step, steps, warmup, barrier):
    """`warmup` untimed steps, then `steps` timed ones between barriers.  Returns the sums over the timed steps."""
    for i in range(warmup):
        dev.render_frames(ubos[i * spp_step:(i + 1) * spp_step])
    barrier()
    out = {"segments": 0, "paths": 0, "kernel_segments": 0, "extend_ms": 0.0, "extend_launches": 0, "device_ms": 0.0}
    ranges = []
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        dev.render_frames(ubos[i * spp_step:(i + 1) * spp_step])
        st = dev.stats()  # waits for the step; counters and HIP-event times of this step
        out["segments"] += st.segments
        out["kernel_segments"] += st.kernel_segments
        out["paths"] += st.paths
        out["extend_ms"] += st.extend_ms
        out["extend_launches"] += st.extend_launches
        out["device_ms"] += st.device_ms
        out["pipeline"] = st.pipeline
        out["stack"] = {"need": int(st.stack_need), "entries_per_lane": int(st.stack_entries),
                        "guarded": bool(st.stack_entries < st.stack_need),
                        "note": "need = the most entries any ray the scene's trees admit could hold; a guarded kernel keeps "
                                "entries_per_lane in LDS and leaves the rare rays that want more to a small launch with "
                                "stacks in memory"}
        ranges.append((int(st.timed_bounce_lo), int(st.timed_bounce_hi)))
    barrier()
    out["elapsed"] = time.perf_counter() - t0
    # the per-segment means of the roofline are those of one range of bounces: every timed step must have run the
    # same launch schedule (the wavefront organisation chooses its hand-over bounce from the previous batch's fills)
    out["timed_bounces"] = ranges[-1]
    out["schedule_stable"] = all(r == ranges[-1] for r in ranges)
    return out


def check_image(image, expect_alpha):
    """Every pixel must have received every sample.  `image`: (H, W, 4), numpy or torch.
    Pixels that are not finite are counted, not refused outright: the reference's own arithmetic produces them,
    rarely (shaders/path_tracing.comp:317-323: a path whose throughput is exactly zero survives roulette when
    rand() returns exactly 0, and `mask *= 1.0f / q` then makes it NaN; seen about once per 1e8 paths in the open
    scene, whose sky shell is black).  More than one pixel in 1e5 would be a fault of the render."""
    alpha = image[..., 3]
    rgb = image[..., :3]
    finite = np.isfinite(rgb).all(axis=2) if isinstance(image, np.ndarray) else rgb.isfinite().all(dim=2)
    n_bad = int((~finite).sum())
    n_pix = int(alpha.shape[0] * alpha.shape[1])
    mean = float(rgb[finite].mean()) if n_bad < n_pix else float("nan")
    return {"alpha_equals_samples": bool((alpha == expect_alpha).all()), "finite": n_bad * 100000 <= n_pix,
            "non_finite_pixels": n_bad, "pixels": n_pix, "samples_per_pixel": float(expect_alpha),
            "alpha_min": float(alpha.min()), "alpha_max": float(alpha.max()), "mean_radiance_of_finite_pixels": mean}


def resolve_scene(L, name, rank):
    """Path of the glTF file of `name` ("outside_synth" is generated: the reference's outside.gltf lacks its buffer)."""
    if name != "outside_synth":
        return name
    from logipathtracer_amd import scenes as _scenes
    sdir = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"lpt_scenes_rank{rank}")
    os.makedirs(sdir, exist_ok=True)
    path = os.path.join(sdir, "outside_synth.gltf")
    if not os.path.exists(path):
        _scenes.write_outside_synth(path, 1.0)
    return path


def describe(wl, spp_step, per_gpu):
    return (f"{os.path.basename(wl['scene_path'])} {wl['width']}x{wl['height']}, {wl['bounces']} bounces, Heitz BSDF, "
            f"{spp_step} spp per step ({per_gpu} per GPU)")


def run_secondary(L, device_index, n_cus, spec):
    """One short single-GPU run of another benchmark configuration on a context of its own."""
    t_begin = time.perf_counter()
    path = resolve_scene(L, spec["scene"], 0)
    wl = {"scene_path": path, "width": spec["width"], "height": spec["height"], "bounces": spec["bounces"]}
    dev = L.Device(device_index)
    try:
        scene = L.Scene(path)
        dev.upload_scene(scene)
        dev.resize(wl["width"], wl["height"])
        dev.set_options(wl["bounces"], L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
        spp, steps, warmup = spec["spp"], spec["steps"], 1
        ubos = L.frame_ubos(scene.camera_world, scene.fov_y, spp * (steps + warmup), seed=0x5EED)
        m = measure(dev, ubos, spp, steps, warmup, dev.synchronize)
        check = check_image(dev.read_accum(), float(spp * (steps + warmup)))
    finally:
        dev.close()
    counters = traversal_counters(scene, ubos, wl["width"], wl["height"], wl["bounces"])
    roof = None
    if m["extend_ms"] > 0:
        roof = roofline(wl, scene, ubos, counters, m["pipeline"], m["kernel_segments"], m["extend_ms"], m["extend_launches"],
                        n_cus, m["timed_bounces"], m.get("stack"))
    return {
        "config": spec["config"], "workload": describe(wl, spp, spp), "value": m["segments"] / m["elapsed"] / 1e6,
        "unit": "Msamples/s", "steps": steps, "warmup": warmup, "ms_per_step": m["elapsed"] * 1e3 / steps,
        "nominal_msamples_per_s": wl["width"] * wl["height"] * spp * steps * wl["bounces"] / m["elapsed"] / 1e6,
        "traced_segments": float(m["segments"]), "camera_paths": float(m["paths"]),
        "pipeline": PIPELINE_NAMES[m["pipeline"]] + " (chosen automatically)", "image_check": check,
        "roofline": roof, "wall_s_including_scene_build": time.perf_counter() - t_begin,
    }


# The other configurations of the benchmark that fit one GPU, each as a short run (BASELINE.json configs 3 to 5;
# the 8-GPU ones appear as the share one GPU renders per step: same resolution, depth and scene, fewer samples).
SECONDARY = [
    {"config": "C3: open scene 1920x1080, 8 bounces", "scene": "outside_synth", "width": 1920, "height": 1080, "bounces": 8,
     "spp": 64, "steps": 2},
    {"config": "C4 resolution: Cornell box 3840x2160, 8 bounces", "scene": CORNELL, "width": 3840, "height": 2160, "bounces": 8,
     "spp": 32, "steps": 2},
    {"config": "C5 resolution and depth: open scene 3840x2160, 16 bounces", "scene": "outside_synth", "width": 3840,
     "height": 2160, "bounces": 16, "spp": 32, "steps": 1},
]


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"--gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        return 2
    import torch
    import torch.distributed as dist
    # LPT_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks then
    # share devices and the gather goes through host memory); the real run uses RCCL ("nccl").
    backend = os.environ.get("LPT_BENCH_BACKEND", "nccl")
    # (a launcher that shows every rank only its own GPU leaves one visible device per process: index 0 then; whether
    # the ranks really sit on different GPUs is checked below by PCI address)
    n_visible = max(1, torch.cuda.device_count())
    device_index = local_rank % n_visible if (backend == "gloo" or local_rank >= n_visible) else local_rank
    torch.cuda.set_device(device_index)
    cuda_dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=cuda_dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import logipathtracer_amd as L
    from logipathtracer_amd import ranks as R
    from logipathtracer_amd.gather import gather_image
    args.scene = resolve_scene(L, args.scene, rank)
    wl = {"scene_path": args.scene, "width": args.width, "height": args.height, "bounces": args.bounces}
    dev = L.Device(device_index)
    scene = L.Scene(args.scene, builder=args.builder, device=dev if args.builder in L.GPU_BUILDERS else None)
    dev.upload_scene(scene)
    dev.resize(args.width, args.height)
    if world > 1:
        dev.set_partition(args.tile, args.tile, rank, world)
    dev.set_options(args.bounces, L.BSDF_HEITZ, args.pipeline, args.frames_in_flight)

    spp_step = args.spp_per_step * world if args.scaling == "weak" else args.spp_per_step
    n_steps_total = args.warmup + args.steps
    ubos = L.frame_ubos(scene.camera_world, scene.fov_y, spp_step * n_steps_total, seed=0x5EED)

    def barrier():
        dev.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    m = measure(dev, ubos, spp_step, args.steps, args.warmup, barrier)
    elapsed = m["elapsed"]
    used_pipeline = m["pipeline"]

    # ---- after the timed region: the image, gathered once over RCCL / xGMI when there are several ranks ----
    expect_alpha = float(spp_step * n_steps_total)
    gather_ms = gather_bytes = None
    if world > 1:
        image, gather_ms, gather_bytes = gather_image(dev, world, args.width, args.height, cuda_dev, backend)
        check = check_image(image, expect_alpha) if rank == 0 else None
    else:
        check = check_image(dev.read_accum(), expect_alpha)
    image_ok = True if check is None else bool(check["alpha_equals_samples"] and check["finite"])

    # ---- who rendered what: one report per rank (rank 0's carries its verdict on the image, so that every rank
    # leaves with the same status) ----
    mine = R.rank_report(rank, local_rank, device_index,
                         {k: m[k] for k in ("device_ms", "extend_ms", "segments", "paths", "elapsed")}, dev.owned_slots())
    mine["image_ok"] = image_ok
    reports = R.collect(mine)
    if not reports[0]["image_ok"]:
        if rank == 0:
            print(f"the image is incomplete or not finite: {check}", file=sys.stderr)
        return 3
    shared = R.shared_devices(reports)
    if shared and backend == "nccl" and world > 1:
        if rank == 0:
            print(f"ranks {shared} rendered on the same GPU: this is not a {world}-GPU run", file=sys.stderr)
        return 4
    elapsed = max(r["elapsed"] for r in reports)  # the slowest rank's time between the barriers
    seg_total = float(sum(r["segments"] for r in reports))
    paths_total = float(sum(r["paths"] for r in reports))

    if rank == 0:
        frames_total = spp_step * args.steps
        nominal = args.width * args.height * frames_total * args.bounces
        cornell = os.path.basename(args.scene) == "cornell_box.gltf"
        per_gpu = spp_step if args.scaling == "strong" else args.spp_per_step
        result = {
            "metric": "Msamples/s (ray segments traced/s) at 1920x1080, 8-bounce Cornell box"
                      if cornell else "Msamples/s (ray segments traced/s)",
            "value": seg_total / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic seeds on the reference's cornell_box.gltf asset; no dataset"
                    if cornell else "synthetic procedural scene (logipathtracer_amd/scenes.py)",
            "config": {
                "workload": describe(wl, spp_step, per_gpu) + (f"; every GPU renders all {spp_step} samples of its own tiles"
                                                              if world > 1 else ""),
                "spp_per_step": spp_step, "tile": args.tile if world > 1 else None,
                "builder": BUILDER_NAMES[args.builder],
                "pipeline": PIPELINE_NAMES[used_pipeline] + (" (chosen automatically)" if args.pipeline == 3 else ""),
                "parallelism": f"image tiles {args.tile}x{args.tile} dealt round-robin to {world} rank(s), scene replicated, "
                               "one all-gather of the tile buffers after the render" if world > 1 else "one GPU",
            },
            "nominal_msamples_per_s": nominal / elapsed / 1e6,
            "traced_segments": seg_total,
            "camera_paths": paths_total,
            "mean_path_length": seg_total / max(paths_total, 1.0),
            "device_ms_rank0": m["device_ms"],
            "extend_ms_rank0": m["extend_ms"],
            "extend_launches_rank0": m["extend_launches"],
            "launch_schedule_stable": m["schedule_stable"],
            "image_check": check,
            "gather_ms": gather_ms,
            "gather_bytes": gather_bytes,
            "backend": backend if world > 1 else None,
            "ranks_seen": len(reports),
            "ranks": reports,
            "distinct_gpus": len({R.device_key(r) for r in reports}),
            "imbalance": R.imbalance(reports) if world > 1 else 1.0,
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N=1 only
            cpu, counters = cpu_baseline(scene, ubos, args.width, args.height, args.bounces, args.cpu_frames)
        else:
            counters = traversal_counters(scene, ubos, args.width, args.height, args.bounces)
        result["cpu_baseline"] = cpu
        result["kernel_segments"] = float(m["kernel_segments"])  # segments traced by the launches extend_ms times
        n_cus = torch.cuda.get_device_properties(device_index).multi_processor_count
        result["roofline"] = roofline(wl, scene, ubos, counters, used_pipeline, m["kernel_segments"], m["extend_ms"],
                                      m["extend_launches"], n_cus, m["timed_bounces"], m.get("stack")) if m["extend_ms"] > 0 else None
        if world == 1 and not args.no_secondary:
            dev.close()  # its queues (tens of GB) are not needed any more
            result["secondary"] = [run_secondary(L, device_index, n_cus, spec) for spec in SECONDARY]
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    dev.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
