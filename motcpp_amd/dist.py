"""Stream sharding and the final track-table gather (the only collective on the path).

Streams are independent, so rank r simply owns streams [r*S, (r+1)*S) with seeds 1234 + global stream id; the tables
are gathered with all_gather_into_tensor on whatever device the tensors live on (cuda -> RCCL over xGMI, cpu -> gloo)."""
import numpy as np


def stream_ids(rank, streams_per_rank):
    return list(range(rank * streams_per_rank, (rank + 1) * streams_per_rank))


def stream_seed(global_stream_id, base=1234):
    return base + int(global_stream_id)


def pack_tables(tables, cap):
    """list of [m_s, 8] arrays -> (padded [S, cap, 8] float32, counts [S] int32)"""
    S = len(tables)
    out = np.zeros((S, cap, 8), np.float32)
    cnt = np.zeros(S, np.int32)
    for s, t in enumerate(tables):
        if t.shape[0] > cap:
            raise ValueError("track table larger than the gather capacity")
        out[s, : t.shape[0]] = t
        cnt[s] = t.shape[0]
    return out, cnt


def gather_tables(padded, counts, device=None):
    """all ranks' padded tables and counts: returns ([world, S, cap, 8], [world, S]) tensors (on `device`)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.as_tensor(padded)
    c = torch.as_tensor(counts)
    if device is not None:
        t, c = t.to(device, non_blocking=True), c.to(device, non_blocking=True)
    if world == 1:
        return t[None], c[None]
    # concatenated output layout (world*S, ...): accepted by both RCCL and gloo
    gt = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    gc = torch.empty((world * c.shape[0],), dtype=c.dtype, device=c.device)
    dist.all_gather_into_tensor(gt, t.contiguous())
    dist.all_gather_into_tensor(gc, c.contiguous())
    return gt.view((world,) + tuple(t.shape)), gc.view(world, -1)


def unpack_tables(gt, gc):
    """inverse of pack+gather: dict global_stream_id -> [m, 8] array"""
    gt, gc = gt.cpu().numpy(), gc.cpu().numpy()
    world, S = gc.shape
    return {r * S + s: gt[r, s, : gc[r, s]].copy() for r in range(world) for s in range(S)}
