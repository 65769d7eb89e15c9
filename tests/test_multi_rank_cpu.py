"""The N>1 path on CPU: two gloo ranks each render the tiles they own (with the oracle standing in for
the kernels), gather their compact tile buffers with one all_gather and assemble the image; the result
must equal the single-rank image bit for bit (seeds depend only on global pixel coordinates)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT

W, H, TILE, BOUNCES, FRAMES = 96, 64, 16, 3, 2


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import logipathtracer_amd as L
    from logipathtracer_amd.partition import Partition
    from oracle import binding as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLDEN, "cornell_scene_binned_sah.npz"))
    osc = O.OracleScene(z["objects"], z["object_bvh"], z["vertices"], z["mesh_bvh"])
    ubos = L.frame_ubos(z["camera_world"], float(z["fov_y"]), FRAMES, seed=99)
    P = Partition(W, H, TILE, TILE, world)
    accum = np.zeros((H, W, 4), dtype=np.float32)
    for tile in P.owned_tiles(rank):
        O.render_frames(osc, ubos, W, H, BOUNCES, tile=tile, accum=accum, threads=1)
    mine = torch.from_numpy(P.pack(accum, rank).reshape(-1))
    gathered = torch.empty(world * mine.numel(), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, mine)  # the one collective of the path
    if rank == 0:
        np.save(out_path, P.assemble(gathered.numpy().reshape(-1, 4)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_rank_image(tmp_path):
    import logipathtracer_amd as L
    from oracle import binding as O
    out = str(tmp_path / "assembled.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(os.path.join(GOLDEN, "cornell_scene_binned_sah.npz"))
    osc = O.OracleScene(z["objects"], z["object_bvh"], z["vertices"], z["mesh_bvh"])
    ubos = L.frame_ubos(z["camera_world"], float(z["fov_y"]), FRAMES, seed=99)
    ref, _ = O.render_frames(osc, ubos, W, H, BOUNCES)
    assert np.array_equal(np.load(out), ref)
