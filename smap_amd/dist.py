"""Frame sharding + result gather for multi-GPU inference (one process per GPU, RCCL over xGMI).

Frames are independent everywhere on the path (eval-mode BN, per-frame association), so
ranks never exchange data while computing.  The only collective is the gather of the
per-frame result records, following the reference's own helper
(lib/utils/comm.py:47-87: pickle, all_gather of byte lengths, then of uint8 buffers padded to the
maximum) and its contiguous per-rank split (lib/utils/dataloader.py:80-85).
"""
import math
import os
import pickle

import torch
import torch.distributed as dist


def shard_range(num_items, world_size, rank):
    """Contiguous block of ceil(N/world) items per rank (dataloader.py:80-85)."""
    per = math.ceil(num_items / world_size) if world_size > 0 else num_items
    st = min(num_items, rank * per)
    return st, min(num_items, st + per)


def _single_rank_shortcut():
    """A world of one rank has nothing to gather -- unless SMAP_FORCE_GATHER=1 asks for the collectives anyway (the `-m gpu`
    test that takes the RCCL path on a one-GPU box: tests/test_entry_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and os.environ.get("SMAP_FORCE_GATHER", "") != "1"


def init_single_rank_group(backend="nccl", device=None):
    """SMAP_FORCE_GATHER=1 without a launcher: a process group of ONE rank on 127.0.0.1, so that the end-of-run gather really
    goes through torch.distributed (backend "nccl" = RCCL) on a one-GPU box."""
    import socket
    if dist.is_initialized():
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=0, world_size=1, **kw)


def gather_bytes(payload, device=None):
    """payload: bytes of this rank.  Returns every rank's bytes, in rank order, on every rank: all_gather of the
    lengths, then of the uint8 buffers padded to the maximum (lib/utils/comm.py:47-87); two host syncs per call (the
    lengths, the payloads), both on the CURRENT stream only.  Works for backend nccl (= RCCL; tensors on `device`)
    and gloo (CPU tensors)."""
    if _single_rank_shortcut():
        return [bytes(payload)]
    world = dist.get_world_size()
    backend = dist.get_backend()
    dev = torch.device(device if device is not None else ("cuda" if backend == "nccl" else "cpu"))
    buf = torch.frombuffer(bytearray(payload) or bytearray(1), dtype=torch.uint8)[:len(payload)].to(dev)
    n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = torch.cat(sizes).cpu().tolist()
    mx = max(max(sizes), 1)
    if buf.numel() < mx:
        buf = torch.cat([buf, torch.zeros(mx - buf.numel(), dtype=torch.uint8, device=dev)])
    outs = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(outs, buf)
    host = torch.stack(outs).cpu().numpy()
    return [host[r, :sz].tobytes() for r, sz in enumerate(sizes)]


def gather_records(records, device=None):
    """Per-rank result records -> the list of every rank's records, in rank order, on every rank.  Serialised with
    pickle, as the reference's helper does (lib/utils/comm.py:57-59): JSON-encoding a batch of poses costs more host time
    than the batch takes on the GPU (8 ms vs 5.5 ms for 8 frames x 8 persons) and JSON is only the FILE format, written
    once by rank 0 after the run (test.py:147-151)."""
    if _single_rank_shortcut():
        return [records]
    return [pickle.loads(b) for b in gather_bytes(pickle.dumps(records, protocol=pickle.HIGHEST_PROTOCOL), device)]


gather_json = gather_records        # earlier name
