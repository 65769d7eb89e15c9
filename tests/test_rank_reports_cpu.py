"""The bookkeeping of a multi-rank run (logipathtracer_amd/ranks.py) over gloo with two ranks: every rank
receives every report, shared GPUs are found, imbalance is largest over smallest."""
import os
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from logipathtracer_amd import ranks as R
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rep = R.rank_report(rank, rank, 0, {"device_ms": 10.0 * (rank + 1), "segments": 100 + rank, "paths": 7, "elapsed": 0.5},
                        owned_slots=4096)
    rep["device"]["pci"] = "0000:%02x:00" % (5 + rank)  # no GPU here: stand-ins for two different cards
    got = R.collect(rep)
    assert [r["rank"] for r in got] == [0, 1] and got[rank]["pid"] == os.getpid()
    assert R.shared_devices(got) == []
    assert R.imbalance(got) == 2.0
    same = [dict(r, device=dict(r["device"], pci="0000:05:00")) for r in got]
    assert R.shared_devices(same) == [[0, 1]]
    if rank == 0:
        open(out_path, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_reports_of_two_ranks(tmp_path):
    out = str(tmp_path / "done")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_a_single_process_is_a_one_rank_run():
    from logipathtracer_amd import ranks as R
    rep = R.rank_report(0, 0, 0, {"device_ms": 1.0, "segments": 1, "paths": 1, "elapsed": 1.0}, owned_slots=64)
    assert R.collect(rep) == [rep] and R.shared_devices([rep]) == [] and R.imbalance([rep]) == 1.0
    # without a PCI address or uuid the device index stands in
    a = dict(rep, device=dict(rep["device"], pci=None, uuid=None))
    b = dict(a, rank=1)
    assert R.shared_devices([a, b]) == [[0, 1]]
