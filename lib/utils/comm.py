"""Process-group helpers with the reference's names (lib/utils/comm.py:12-87) on top of smap_amd.dist.

`test.py` of the reference imports `is_main_process` from here; `all_gather(data)` is the variable-length gather of
picklable objects it uses for results.  `reduce_dict` (loss averaging, comm.py:90-117) belongs to training."""
import torch.distributed as dist

from smap_amd.dist import gather_records


def _active():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _active() else 1


def get_rank():
    return dist.get_rank() if _active() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    """Barrier over all ranks; nothing to do in a single process."""
    if _active() and dist.get_world_size() > 1:
        dist.barrier()


def all_gather(data, device=None):
    """list[data from each rank], in rank order, on every rank (any picklable object; RCCL or gloo)."""
    return gather_records(data, device)
