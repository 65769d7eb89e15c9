#!/usr/bin/env python3
"""Benchmark of the RendererPT compute path on MI355X.

Metric (BASELINE.json): Msamples/s at 1920x1080, 8-bounce Cornell box, where a sample is one
ray segment actually traced (one iteration of traceRay's loop, shaders/path_tracing.comp:216-324).
`value` counts traced segments only; the nominal figure W*H*spp*bounces/t is reported beside it.

A step renders `spp_per_step` frames (one sample per pixel each) of the whole image; with N GPUs the
image is cut into tiles dealt round-robin to the ranks and spp_per_step grows with N, so each GPU
traces the same number of camera paths per step at every N (weak scaling).  No collective runs in
the timed region; the single gather of the HDR tile buffers happens after it and is timed apart.

  python bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec
# Vector L1 (the path every per-lane gather of the search takes): 64 B per clock and CU.
L1_BYTES_PER_CLK_PER_CU = 64.0
PEAK_CLOCK_GHZ = 2.4  # MI355X peak engine clock (what hipDeviceProp_t::clockRate reports on this part)


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
    ap.add_argument("--spp-per-step", type=int, default=64, help="frames per step and per GPU")
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "cornell_box.gltf"))
    ap.add_argument("--builder", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=3, help="0 wavefront, 1 megakernel, 2 persistent, 3 automatic, 4 hybrid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample (0: automatic)")
    return ap.parse_args()


def cpu_baseline(scene, ubos, width, height, bounces, frames):
    """Times the CPU oracle (OpenMP over pixels) on a bounded sample of the same workload and returns
    (baseline dict, oracle work counters of the sample)."""
    from oracle import binding as O
    osc = O.OracleScene(scene.objects, scene.object_bvh, scene.vertices, scene.mesh_bvh, getattr(scene, "textures", ()))
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    if frames <= 0:
        t0 = time.perf_counter()
        O.render_frames(osc, ubos[:1], width, height, bounces, threads=threads)
        one = time.perf_counter() - t0
        frames = int(max(1, min(len(ubos), 15.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    _, st = O.render_frames(osc, ubos[:frames], width, height, bounces, threads=threads)
    dt = time.perf_counter() - t0
    base = {"value": st["segments"] / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{frames} frame(s) of {width}x{height}, {bounces} bounces, same scene and seeds "
                      f"({st['segments']} segments in {dt:.2f} s)"}
    return base, st


def traversal_counters(scene, ubos, width, height, bounces):
    """Oracle work counters (nodes and triangles per segment) from one frame at a quarter of the
    resolution: enough for the per-segment means the roofline needs, in well under a second."""
    from oracle import binding as O
    osc = O.OracleScene(scene.objects, scene.object_bvh, scene.vertices, scene.mesh_bvh, getattr(scene, "textures", ()))
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    _, st = O.render_frames(osc, ubos[:1], max(16, width // 4), max(16, height // 4), bounces, threads=threads)
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


def roofline(args, scene, ubos, counters, used_pipeline, kernel_seg, ext_ms, ext_launches, props, timed_bounces):
    """Roofline block of the dominant kernel: k_extend (wavefront organisation) or k_trace.

    What bounds these kernels is the vector L1: every search step is a per-lane gather of one small record
    (64-byte node, 48-byte triangle, 64-byte object record) in 16-byte loads at unrelated addresses, and the
    scene is cache-resident, so hardly any of it reaches HBM.  The block therefore prices the kernel against
    the L1's nominal rate (64 B per clock and CU):
      achieved = 16 B x the per-lane loads the kernel issues per traced segment x segments / kernel time,
      loads per segment = 4 per inner node visited + 3 per triangle tested + 4 per object entered
                          + the ray/queue/surface records of the kernel (below),
    with the per-segment means taken from the CPU oracle's counters over the bounces the kernel traces and
    the kernel time measured live with HIP events.  Beside it: the measured ceiling of such gathers on this
    chip (profiles/r02_gather_peak.json, tools/gather_peak.hip), the algorithmic bytes in interchange terms
    (48 B per node fetched + 36 B per triangle + state), and HBM traffic from the committed PMC passes."""
    kernel = {0: "k_extend", 2: "k_trace", 4: "k_trace"}.get(used_pipeline, "k_megakernel")
    # the bounces whose segments the timed kernel traces: [lo, hi) (lpt_stats; the wavefront organisation hands
    # the late bounces to one persistent launch where the queues become short)
    lo, hi = timed_bounces
    upto_hi = counters if hi == args.bounces else traversal_counters(scene, ubos, args.width, args.height, hi)
    upto_lo = traversal_counters(scene, ubos, args.width, args.height, lo) if lo > 0 else None
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
    achieved = 16.0 * loads_per_seg * kernel_seg / secs / 1e9
    clock_ghz = PEAK_CLOCK_GHZ
    cus = props.multi_processor_count
    peak = cus * L1_BYTES_PER_CLK_PER_CU * clock_ghz  # GB/s
    bytes_per_seg = 48.0 * n_node + 36.0 * n_tri + state_bytes
    out = {
        "kernel": kernel, "bound": "l1", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
        "traffic": None,
        "note": f"achieved = 16 B x per-lane loads ({loads_per_seg:.1f} per segment: 4 per inner node, 3 per triangle, 4 per "
                f"object entered, {loads_note}) x the segments {kernel} traced (bounces {lo} to {hi - 1}) / its time "
                f"(HIP events); peak = {cus} CUs x 64 B/clk x {clock_ghz:.2f} GHz, the vector L1's nominal rate. Random "
                "16-byte-granular gathers reach far less than that (gather_ceiling); HBM is not the bound: the scene is "
                "cache-resident (hbm_from_profile)",
        "loads_per_segment": loads_per_seg, "inner_nodes_per_segment": n_inner, "tris_per_segment": n_tri,
        "objects_entered_per_segment": n_obj, "hits_per_segment": p_hit,
        "avg_launch_ms": ext_ms / max(ext_launches, 1), "launches": ext_launches,
        "segments_per_launch": kernel_seg / max(ext_launches, 1),
        # algorithmic bytes in interchange terms (48 B per node fetched incl. both tested children, 36 B per triangle)
        "algorithmic_bytes_per_segment": bytes_per_seg,
        "algorithmic_gbs": bytes_per_seg * kernel_seg / secs / 1e9,
        "algorithmic_note": f"48 B/node fetched + 36 B/triangle + {state_note}; exceeds the HBM peak because the scene is "
                            "served from L1/L2",
    }
    gpath = os.path.join(ROOT, "profiles", "r02_gather_peak.json")
    if os.path.exists(gpath):
        try:
            g = json.load(open(gpath))["ceilings_gbs"]
            out["gather_ceiling"] = {
                "l1_resident_gbs": g["records_64B_l1_resident_16KiB_table"],
                "l2_resident_gbs": g["records_64B_l2_resident_2MiB_table"],
                "frac_of_l1_resident": achieved / g["records_64B_l1_resident_16KiB_table"],
                "source": "profiles/r02_gather_peak.json (random 64-byte records, every lane its own; lanes of the search "
                          "that meet at one node share a line, so the kernel may exceed the random-address ceiling)",
            }
        except Exception:
            pass
    # HBM traffic and unit-busy fractions of this workload from the committed rocprofv3 PMC passes (not measured in this run)
    tpath = os.path.join(ROOT, "profiles", {"k_extend": "extend_traffic.json", "k_trace": "trace_traffic.json"}.get(kernel, "none"))
    default_workload = (os.path.basename(args.scene) == "cornell_box.gltf" and args.bounces == 8
                        and (args.width, args.height) == (1920, 1080))
    if os.path.exists(tpath) and default_workload:
        try:
            t = json.load(open(tpath))
            if t.get("pipeline", used_pipeline) == used_pipeline:
                per_seg = t["hbm_bytes_per_launch"] / t["segments_per_launch"]
                gbs = per_seg * kernel_seg / secs / 1e9
                out["hbm_from_profile"] = {
                    "bytes_per_segment": per_seg, "bytes_per_launch": per_seg * kernel_seg / max(ext_launches, 1),
                    "gbs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "peak_gbs": HBM_PEAK_GBS,
                    "unit_busy": t.get("units"), "source": "profiles/" + os.path.basename(tpath),
                    "rule": t.get("rule"),
                }
        except Exception:
            pass
    return out


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
    device_index = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    cuda_dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=cuda_dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import logipathtracer_amd as L
    from logipathtracer_amd.gather import gather_image
    if args.scene == "outside_synth":
        # stand-in for the reference's outside.gltf (its binary buffer is not in the snapshot)
        from logipathtracer_amd import scenes as _scenes
        sdir = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"lpt_scenes_rank{rank}")
        os.makedirs(sdir, exist_ok=True)
        args.scene = os.path.join(sdir, "outside_synth.gltf")
        _scenes.write_outside_synth(args.scene, 1.0)
    dev = L.Device(device_index)
    scene = L.Scene(args.scene, builder=args.builder, device=dev if args.builder in L.GPU_BUILDERS else None)
    dev.upload_scene(scene)
    dev.resize(args.width, args.height)
    if world > 1:
        dev.set_partition(args.tile, args.tile, rank, world)
    dev.set_options(args.bounces, L.BSDF_HEITZ, args.pipeline, args.frames_in_flight)

    spp_step = args.spp_per_step * world
    n_steps_total = args.warmup + args.steps
    ubos = L.frame_ubos(scene.camera_world, scene.fov_y, spp_step * n_steps_total, seed=0x5EED)

    def barrier():
        dev.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_stats = []

    def run_step(i):
        dev.render_frames(ubos[i * spp_step:(i + 1) * spp_step])

    for i in range(args.warmup):
        run_step(i)
    barrier()
    t0 = time.perf_counter()
    seg = paths = kernel_seg_total = 0
    ext_ms = 0.0
    ext_launches = 0
    dev_ms = 0.0
    for i in range(args.warmup, n_steps_total):
        run_step(i)
        st = dev.stats()  # waits for the step; counters and HIP-event times of this step
        seg += st.segments
        kernel_seg_total += st.kernel_segments
        paths += st.paths
        ext_ms += st.extend_ms
        ext_launches += st.extend_launches
        dev_ms += st.device_ms
        used_pipeline = st.pipeline
        timed_bounces = (int(st.timed_bounce_lo), int(st.timed_bounce_hi))
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- after the timed region: gather the HDR tile buffers once (RCCL over xGMI) ----
    gather_ms = None
    if world > 1:
        image, gather_ms = gather_image(dev, world, args.width, args.height, cuda_dev, backend)
        if rank == 0:
            # every pixel must have received every sample of the run, from whichever rank owns it
            alpha = image[..., 3]
            expect = float(spp_step * n_steps_total)
            if not bool((alpha == expect).all()):
                print(f"gathered image is incomplete: alpha in [{float(alpha.min())}, {float(alpha.max())}], "
                      f"expected {expect}", file=sys.stderr)
                return 3

    if world > 1:
        t = torch.tensor([elapsed, float(seg), float(paths), ext_ms, float(ext_launches)], dtype=torch.float64,
                         device=cuda_dev if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        seg_total, paths_total = float(tsum[1]), float(tsum[2])
    else:
        seg_total, paths_total = float(seg), float(paths)

    if rank == 0:
        frames_total = spp_step * args.steps
        nominal = args.width * args.height * frames_total * args.bounces
        result = {
            "metric": "Msamples/s (ray segments traced/s) at 1920x1080, 8-bounce Cornell box"
                      if "cornell_box" in args.scene else "Msamples/s (ray segments traced/s)",
            "value": seg_total / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic seeds on the reference's cornell_box.gltf asset; no dataset"
                    if "cornell_box" in args.scene else "synthetic procedural scene (logipathtracer_amd/scenes.py)",
            "config": {
                "workload": f"{os.path.basename(args.scene)} {args.width}x{args.height}, {args.bounces} bounces, "
                            f"Heitz BSDF, {spp_step} spp per step ({args.spp_per_step} per GPU)",
                "spp_per_step": spp_step, "tile": args.tile if world > 1 else None,
                "builder": ["split_bvh", "binned_sah", "lbvh_gpu", "ploc_gpu"][args.builder],
                "pipeline": ["wavefront", "megakernel", "persistent", "auto", "hybrid"][used_pipeline]
                            + (" (chosen automatically)" if args.pipeline == 3 else ""),
            },
            "nominal_msamples_per_s": nominal / elapsed / 1e6,
            "traced_segments": seg_total,
            "camera_paths": paths_total,
            "mean_path_length": seg_total / max(paths_total, 1.0),
            "device_ms_rank0": dev_ms,
            "extend_ms_rank0": ext_ms,
            "extend_launches_rank0": ext_launches,
            "gather_ms": gather_ms,
        }
        cpu = None
        counters = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N=1 only
            cpu, counters = cpu_baseline(scene, ubos, args.width, args.height, args.bounces, args.cpu_frames)
        else:
            counters = traversal_counters(scene, ubos, args.width, args.height, args.bounces)
        result["cpu_baseline"] = cpu
        result["kernel_segments"] = float(kernel_seg_total)  # segments traced by the launches extend_ms times
        result["roofline"] = roofline(args, scene, ubos, counters, used_pipeline, kernel_seg_total, ext_ms, ext_launches,
                                      torch.cuda.get_device_properties(device_index), timed_bounces) if counters and ext_ms > 0 else None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    dev.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
