"""Frame sharding + result gather for multi-GPU inference (one process per GPU, RCCL over xGMI).

Frames are independent everywhere on the path (eval-mode BN, per-frame association), so
ranks never exchange data while computing.  The only collective is the gather of the
per-frame result records, following the reference's own helper
(lib/utils/comm.py:47-87: all_gather of byte lengths, then of uint8 buffers padded to the
maximum) and its contiguous per-rank split (lib/utils/dataloader.py:80-85).
"""
import json
import math

import torch
import torch.distributed as dist


def shard_range(num_items, world_size, rank):
    """Contiguous block of ceil(N/world) items per rank (dataloader.py:80-85)."""
    per = math.ceil(num_items / world_size) if world_size > 0 else num_items
    st = min(num_items, rank * per)
    return st, min(num_items, st + per)


def gather_json(records, device=None):
    """records: JSON-serialisable object of this rank.  Returns the list of every rank's
    object, in rank order, on every rank.  Works for backend nccl (=RCCL; tensors on `device`)
    and gloo (CPU tensors)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [records]
    world = dist.get_world_size()
    backend = dist.get_backend()
    dev = torch.device(device if device is not None else ("cuda" if backend == "nccl" else "cpu"))
    payload = json.dumps(records, separators=(",", ":")).encode()
    buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    n = torch.tensor([buf.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if buf.numel() < mx:
        buf = torch.cat([buf, torch.zeros(mx - buf.numel(), dtype=torch.uint8, device=dev)])
    outs = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [json.loads(bytes(o[:s].cpu().numpy().tobytes()).decode()) for o, s in zip(outs, sizes)]
