"""Bookkeeping of a multi-rank run: who took part, on which GPU, and how evenly the work was spread.

The render itself needs no exchange between ranks (tiles are independent, shaders/path_tracing.comp:341 seeds
a path from its global pixel alone); what is collected here is evidence about the run: one report per rank,
gathered with `all_gather_object`, so that a benchmark line can show that N distinct GPUs did the work and
what each of them contributed.  Works with any backend (the CPU tests use gloo).
"""
import os
import socket


def device_identity(device_index):
    """What identifies the GPU a rank renders on: PCI address and uuid as the runtime reports them (None where
    this torch build does not expose a field)."""
    import torch
    ident = {"device_index": int(device_index), "host": socket.gethostname(), "name": None, "uuid": None, "pci": None}
    try:
        p = torch.cuda.get_device_properties(device_index)
    except Exception:
        return ident
    ident["name"] = getattr(p, "name", None)
    uuid = getattr(p, "uuid", None)
    ident["uuid"] = str(uuid) if uuid is not None else None
    dom, bus, dev = (getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if bus is not None:
        ident["pci"] = "%04x:%02x:%02x" % (int(dom or 0), int(bus), int(dev or 0))
    return ident


def device_key(report):
    """Two reports with the same key rendered on the same GPU."""
    d = report["device"]
    unique = d.get("pci") or d.get("uuid")
    return (d.get("host"), unique if unique else "index %d" % d["device_index"])


def collect(report):
    """Every rank passes its own report (a dict); every rank receives the list of all reports, by rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [report]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, report)
    return out


def shared_devices(reports):
    """Groups of ranks that rendered on one and the same GPU (empty when every rank had its own)."""
    by_key = {}
    for r in reports:
        by_key.setdefault(device_key(r), []).append(r["rank"])
    return [ranks for ranks in by_key.values() if len(ranks) > 1]


def imbalance(reports, field="device_ms"):
    """Largest over smallest per-rank value of `field`: 1.0 is a perfectly even split."""
    vals = [float(r[field]) for r in reports]
    lo, hi = min(vals), max(vals)
    return hi / lo if lo > 0 else float("inf")


def rank_report(rank, local_rank, device_index, stats_sum, owned_slots):
    """`stats_sum`: dict with device_ms, extend_ms, segments, paths summed over the timed steps of this rank."""
    rep = {"rank": int(rank), "local_rank": int(local_rank), "pid": os.getpid(), "device": device_identity(device_index),
           "owned_slots": int(owned_slots)}
    rep.update({k: float(v) for k, v in stats_sum.items()})
    return rep
