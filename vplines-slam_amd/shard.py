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


# ---- results of a sharded batch: one all-gather of the per-window states at batch end --------------
STATE_DOUBLES = 11 * 16 + 7     # per window: 11 x (pose 7 + speed/bias 9) + extrinsic 7 = 183 doubles


def pack_states(windows):
    """[n, 183] float64: pose, speed_bias, ex_pose of every window (what a consumer of the batch needs back)"""
    import numpy as np
    out = np.zeros((len(windows), STATE_DOUBLES))
    for i, w in enumerate(windows):
        out[i, :77] = np.asarray(w.pose).reshape(-1)
        out[i, 77:176] = np.asarray(w.speed_bias).reshape(-1)
        out[i, 176:] = np.asarray(w.ex_pose).reshape(-1)
    return out


def unpack_state(row):
    """(pose [11,7], speed_bias [11,9], ex_pose [7]) of one gathered row"""
    return row[:77].reshape(11, 7), row[77:176].reshape(11, 9), row[176:183]


def gather_states(dist, local_states, total_windows, device=None):
    """All-gather of the block-partitioned per-window states (split_batch order) -> [total_windows, 183] on every rank.
    dist=None: single process, returns the local array.  Blocks of unequal size are padded to the largest one."""
    import numpy as np
    import torch
    if dist is None:
        assert len(local_states) == total_windows
        return np.asarray(local_states)
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [split_batch(total_windows, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros((cap, STATE_DOUBLES), dtype=torch.float64, device=device)
    lo, hi = sizes[rank]
    assert len(local_states) == hi - lo
    if hi > lo:
        buf[: hi - lo] = torch.as_tensor(np.asarray(local_states), dtype=torch.float64).to(buf.device)
    out = torch.zeros((world * cap, STATE_DOUBLES), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, cap, STATE_DOUBLES)
    return np.concatenate([out[r, : sizes[r][1] - sizes[r][0]] for r in range(world)], axis=0)


def gather_states_device(dist, ctx, n_local, total_windows, device):
    """The same all-gather with the local block packed ON THE DEVICE (vpl_ba_pack_states_device) and handed to the collective
    as a device tensor: no host staging on the way in.  Returns the gathered [total_windows, 183] DEVICE tensor (window
    order).  dist=None: the local block alone."""
    import torch
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    sizes = [split_batch(total_windows, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    assert n_local == sizes[rank][1] - sizes[rank][0]
    buf = torch.zeros((cap, STATE_DOUBLES), dtype=torch.float64, device=device)
    # the zero fill runs on torch's current stream, the pack on the context's: order them explicitly (with a non-default torch
    # stream or a non-blocking context stream nothing else does -- ADVICE r3)
    torch.cuda.current_stream(device).synchronize()
    if n_local:
        ctx.pack_states_device(n_local, buf.data_ptr())
        ctx.synchronize()
    if dist is None:
        return buf[:n_local]
    out = torch.empty((world * cap, STATE_DOUBLES), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.view(world, cap, STATE_DOUBLES)
    if all(hi - lo == cap for lo, hi in sizes):
        return out.reshape(world * cap, STATE_DOUBLES)
    return torch.cat([out[r, : sizes[r][1] - sizes[r][0]] for r in range(world)], dim=0)


def reduce_max_vec(dist, values, device=None):
    """element-wise MAX over ranks of a short list of floats (parity / timing summary)"""
    import torch
    if dist is None:
        return [float(x) for x in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.cpu()]
