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


# ---- packed tables (mot_bt_step_packed): rows of all streams back to back + counts ---------------------------------------
class _DeviceArray:
    """zero-copy view of device memory for torch.as_tensor (CUDA array interface; works for HIP memory on ROCm)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr, shape, dtype, device):
    """torch tensor over `ptr` (device memory owned by the library) — no copy."""
    import torch
    typestr = {"float32": "<f4", "int32": "<i4"}[str(dtype).replace("torch.", "")]
    return torch.as_tensor(_DeviceArray(ptr, shape, typestr), device=device)


def gather_packed(rows_list, counts_list, rows_cap):
    """Gather of packed tables without leaving the device they are on.
    rows_list: per sub-batch tensors [n_p, 8] (the emitted rows, stream after stream); counts_list: per sub-batch [S_p] int32.
    They are laid into one fixed-size send buffer [rows_cap, 8] (+ counts [S]) — device to device when the inputs are device
    tensors, so nothing bounces through the host — and exchanged with all_gather_into_tensor (RCCL over xGMI on cuda tensors,
    gloo on cpu tensors). Returns ([world, rows_cap, 8], [world, S]): rank r's stream s starts at counts[r, :s].sum()."""
    import torch
    import torch.distributed as dist
    dev = rows_list[0].device
    send = torch.zeros((rows_cap, 8), dtype=torch.float32, device=dev)
    o = 0
    for r in rows_list:
        n = int(r.shape[0])
        if o + n > rows_cap:
            raise ValueError("packed tables larger than the gather capacity")
        send[o:o + n].copy_(r, non_blocking=True)
        o += n
    cnt = torch.cat([c.to(dev) for c in counts_list]).to(torch.int32)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return send[None], cnt[None]
    gt = torch.empty((world * rows_cap, 8), dtype=torch.float32, device=dev)
    gc = torch.empty((world * cnt.shape[0],), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gt, send)
    dist.all_gather_into_tensor(gc, cnt)
    return gt.view(world, rows_cap, 8), gc.view(world, -1)


def unpack_packed(gt, gc):
    """dict global_stream_id -> [m, 8] array from gather_packed's result"""
    gt, gc = gt.cpu().numpy(), gc.cpu().numpy()
    world, S = gc.shape
    out = {}
    for r in range(world):
        off = np.concatenate([[0], np.cumsum(gc[r])])
        for s in range(S):
            out[r * S + s] = gt[r, off[s]:off[s + 1]].copy()
    return out
