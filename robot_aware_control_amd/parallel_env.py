"""When do the collectives of the path run?  One process per GPU under `torch.distributed` (backend "nccl" = RCCL over xGMI).

`active()` is true with an initialised process group of more than one rank -- and, with RAC_DIST_FORCE=1, of ONE rank
too: every collective call site of the path (the planner's candidate broadcast and cost all-gather, the trainer's
parameter broadcast and gradient all-reduce / reduce-scatter) then executes on a one-GPU box through RCCL itself
(tests/test_gpu_nccl.py), where two ranks cannot share a device under RCCL."""
from __future__ import annotations

import os

import torch.distributed as dist


def active() -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("RAC_DIST_FORCE", "0") == "1"


def world_rank():
    """(world size, rank) for sharding decisions: (1, 0) when the collectives are off."""
    return (dist.get_world_size(), dist.get_rank()) if active() else (1, 0)
