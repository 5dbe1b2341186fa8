"""Batch sharding of independent sliding windows over the GPUs of one node.

The path shards by construction: a window solve never reads another window.  Every rank owns a
contiguous block of global window indices, derives its inputs from the window index alone (the
synthetic generator is seeded per window) and runs the whole solve locally; there is no collective
in the data path.  The only exchanges are (i) the MAX of the timed region over ranks and (ii) an
all-gather of a few per-rank scalars (parity / iteration statistics) at the end of a run."""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def window_range(rank, world, windows_per_gpu):
    """weak scaling: every rank solves `windows_per_gpu` windows; global ids are contiguous per rank"""
    lo = rank * windows_per_gpu
    return lo, lo + windows_per_gpu


def split_batch(total_windows, rank, world):
    """strong scaling variant: block partition of a fixed batch ([g*B/G, (g+1)*B/G) )"""
    lo = (total_windows * rank) // world
    hi = (total_windows * (rank + 1)) // world
    return lo, hi


def window_seeds(config_id, global_index):
    """(seed of the preceding window A, seed of the timed window B, trajectory phase) of one window"""
    from . import workload
    return (workload.seed_for(config_id, 2 * global_index), workload.seed_for(config_id, 2 * global_index + 1),
            0.37 * global_index)


def reduce_max(dist, value, device=None):
    """MAX over ranks of a python float (the bench contract's elapsed time)"""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stats(dist, stats, device=None):
    """all-gather of a small fixed-length list of floats per rank -> list of lists"""
    import torch
    t = torch.tensor(list(stats), dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]
