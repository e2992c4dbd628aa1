"""Sharded grid mapping: scans are split across ranks, every rank
inserts its share into a private count grid, and the two integer count layers are summed with
one all-reduce at the end.  Counts are additive integers, so the sharded map equals the map of
the single-process insertion exactly (CPMap::updateMap, src/drawmap/PMap.cpp:46-91).

`update_fn(first, count) -> (hit, visits)` is the per-rank insertion: on a GPU box it wraps
capi.Grid.update_batch + device_ptrs; the CPU test passes the oracle.  This module is exchange
plumbing only (torch.distributed: RCCL on the GPUs, gloo in the CPU tests).
"""
import torch
import torch.distributed as dist

from .pf import shard_bounds


def sharded_grid_update(n_scans, update_fn):
    """Returns the summed (hit, visits) int32 layers of all ranks' shares of `n_scans` scans."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    first, count = shard_bounds(n_scans, rank, world)
    hit, visits = update_fn(first, count)
    hit = hit.to(torch.int32).contiguous()
    visits = visits.to(torch.int32).contiguous()
    if world > 1:
        # one collective for both layers
        both = torch.stack([hit, visits])
        dist.all_reduce(both, op=dist.ReduceOp.SUM)
        hit, visits = both[0], both[1]
    return hit, visits
