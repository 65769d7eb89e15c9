"""The one collective of a multi-GPU render: an all-gather of every rank's compact tile buffer.

Each rank accumulates its tiles in the compact buffer lpt_owned_accum_device points at (equal size on
every rank).  After rendering, one `all_gather_into_tensor` (RCCL over xGMI with the "nccl" backend)
delivers all buffers to every rank and lpt_assemble_gathered scatters them into a scan-line image.
torch is used for the process group and for device memory only.
"""
import time


class _DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 3}


def owned_accum_tensor(dev, torch_device):
    """The context's compact accumulation buffer as a float32 torch tensor that aliases it."""
    import torch
    n_floats = dev.owned_slots() * 4
    return torch.as_tensor(_DeviceArray(dev.owned_accum_device(), n_floats), device=torch_device)


def gather_image(dev, world, width, height, torch_device, backend="nccl"):
    """All ranks call this after rendering.  Returns (image, milliseconds of the collective, bytes every rank
    received): `image` is a (height, width, 4) float32 device tensor holding every rank's pixels (on every rank).
    `backend` "nccl" gathers device to device; "gloo" (rehearsals on fewer GPUs than ranks) goes through
    host memory."""
    import torch
    import torch.distributed as dist
    dev.synchronize()
    # The collective reads the context's compact buffer in place: the tensor below aliases the device memory
    # lpt_owned_accum_device points at (plain hipMalloc memory, which RCCL takes like any other device pointer).
    mine = owned_accum_tensor(dev, torch_device)
    n_floats = mine.numel()
    gathered = torch.empty(world * n_floats, dtype=torch.float32, device=torch_device)
    torch.cuda.synchronize(torch_device)
    dist.barrier()
    t0 = time.perf_counter()
    if backend == "nccl":
        dist.all_gather_into_tensor(gathered, mine)
    else:
        host = torch.empty(world * n_floats, dtype=torch.float32)
        dist.all_gather_into_tensor(host, mine.cpu())
        gathered.copy_(host)
    torch.cuda.synchronize(torch_device)
    ms = (time.perf_counter() - t0) * 1e3
    image = torch.empty(height * width * 4, dtype=torch.float32, device=torch_device)
    dev.assemble_gathered(gathered.data_ptr(), world, image.data_ptr())
    dev.synchronize()
    return image.view(height, width, 4), ms, world * n_floats * 4
