"""world_size-2 gloo test (CPU) of the N>1 path: stream sharding (disjoint seeds) and the final track-table gather.
The tracker work itself needs a GPU; here each rank fabricates deterministic tables for its streams."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from motcpp_amd import dist as mdist

S, CAP = 3, 16


def fake_table(gid):
    r = np.random.default_rng(gid)
    m = int(r.integers(0, CAP))
    t = r.uniform(0, 100, (m, 8)).astype(np.float32)
    t[:, 4] = np.arange(1, m + 1) + 1000 * gid
    return t


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = mdist.stream_ids(rank, S)
    padded, cnt = mdist.pack_tables([fake_table(g) for g in ids], CAP)
    gt, gc = mdist.gather_tables(padded, cnt)
    got = mdist.unpack_tables(gt, gc)
    ok = set(got) == set(range(world * S)) and all(np.array_equal(got[g], fake_table(g)) for g in got)
    # the packed form (what mot_bt_step_packed leaves on the device): two "sub-batches" of rows back to back + counts
    tabs = [fake_table(g) for g in ids]
    halves = [tabs[:2], tabs[2:]]
    rows_list = [torch.from_numpy(np.concatenate(h) if h else np.zeros((0, 8), np.float32)) for h in halves]
    counts_list = [torch.tensor([t.shape[0] for t in h], dtype=torch.int32) for h in halves]
    pt, pc = mdist.gather_packed(rows_list, counts_list, S * CAP)
    got2 = mdist.unpack_packed(pt, pc)
    ok = ok and set(got2) == set(range(world * S)) and all(np.array_equal(got2[g], fake_table(g)) for g in got2)
    # same barrier + max-over-ranks timing pattern as bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    q.put((rank, ok, [mdist.stream_seed(g) for g in ids]))
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    seeds = sorted(sd for _, _, ss in res for sd in ss)
    assert seeds == [1234 + i for i in range(2 * S)]  # disjoint, contiguous seeds across ranks


def test_pack_roundtrip_single_rank():
    tables = [fake_table(g) for g in range(4)]
    padded, cnt = mdist.pack_tables(tables, CAP)
    gt, gc = mdist.gather_tables(padded, cnt)
    got = mdist.unpack_tables(gt, gc)
    assert all(np.array_equal(got[g], tables[g]) for g in range(4))
