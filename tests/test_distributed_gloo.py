"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): the batch sharding used by bench.py,
the MAX-over-ranks timing reduction and the statistics all-gather.  The data path itself has no
collective (independent windows), so what can be tested without GPUs is exactly this plumbing."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nw, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import vplines_slam_amd as v
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, w, lr = v.shard.rank_world()
    lo, hi = v.shard.window_range(r, w, nw)
    seeds = [v.shard.window_seeds(3, g) for g in range(lo, hi)]
    # every rank generates ITS windows only; checksums of the generated inputs are exchanged
    cfg = v.workload.config(12, 4, True)
    csum = 0.0
    for (sa, sb, t) in seeds:
        wnd = v.workload.generate(sb, cfg, t + cfg.kf_dt)
        csum += float(wnd.pose.sum() + wnd.point_obs.sum() + wnd.line_obs.sum())
    elapsed = v.shard.reduce_max(dist, 1.0 + rank)           # rank-dependent fake timings
    stats = v.shard.gather_stats(dist, [float(lo), float(hi), csum])
    slo, shi = v.shard.split_batch(7, r, w)
    # results of a block-partitioned batch of 7 windows (uneven blocks 3 + 4): every rank contributes the states of ITS
    # windows, one all-gather puts the whole batch on every rank in window order (bench.py does this over RCCL)
    import numpy as np
    local = []
    for g in range(slo, shi):
        wnd = v.workload.generate(v.shard.window_seeds(3, g)[1], cfg, 0.0)
        local.append(wnd)
    gathered = v.shard.gather_states(dist, v.shard.pack_states(local), 7)
    par = v.shard.reduce_max_vec(dist, [float(rank), 10.0 - rank])
    out_q.put((rank, elapsed, stats, (slo, shi), [s[1] for s in seeds], gathered, par))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    world, nw = 2, 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # MAX over ranks
    assert all(abs(r[1] - 2.0) < 1e-12 for r in res)
    # both ranks see the same gathered table; ranges are disjoint and contiguous
    assert res[0][2] == res[1][2]
    table = res[0][2]
    assert [int(table[0][0]), int(table[0][1]), int(table[1][0]), int(table[1][1])] == [0, 3, 3, 6]
    assert table[0][2] != table[1][2]            # different windows on different ranks
    # seeds are disjoint across ranks and deterministic
    assert not set(res[0][4]) & set(res[1][4])
    # strong-scaling split covers the batch exactly
    assert res[0][3] == (0, 3) and res[1][3] == (3, 7)
    # the gathered result set is identical on both ranks, in window order, and equals what one process computes
    import numpy as np
    sys.path.insert(0, ROOT)
    import vplines_slam_amd as v
    g0, g1 = res[0][5], res[1][5]
    assert g0.shape == (7, v.shard.STATE_DOUBLES) and np.array_equal(g0, g1)
    cfg = v.workload.config(12, 4, True)
    ref = v.shard.pack_states([v.workload.generate(v.shard.window_seeds(3, g)[1], cfg, 0.0) for g in range(7)])
    assert np.array_equal(g0, ref)
    pose, sb, ex = v.shard.unpack_state(g0[4])
    assert pose.shape == (11, 7) and sb.shape == (11, 9) and ex.shape == (7,)
    assert res[0][6] == [1.0, 10.0] and res[1][6] == [1.0, 10.0]     # element-wise MAX over ranks


def test_split_batch_covers_everything():
    sys.path.insert(0, ROOT)
    import vplines_slam_amd as v
    for total in (1, 7, 512):
        for world in (1, 2, 4, 8):
            ranges = [v.shard.split_batch(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
