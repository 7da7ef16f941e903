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


# ---- native gather (mot_comm_*: RCCL called from the library on the context's stream) ------------------------------------
class NativeComm:
    """One RCCL communicator bound to a motcpp_amd Context. The 128-byte unique id travels through `exchange`, a callable
    that takes rank 0's bytes (None on the other ranks) and returns them on every rank — e.g. a torch.distributed broadcast
    (default when a process group is initialised), MPI, or a file. World size 1 needs no exchange."""

    def __init__(self, ctx, world=None, rank=None, exchange=None):
        import ctypes as C
        self.ctx, self.lib = ctx, ctx.lib
        if world is None:
            import torch.distributed as dist
            world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
        self.world, self.rank = int(world), int(rank)
        buf = (C.c_char * 128)()
        if self.rank == 0:
            if self.lib.mot_comm_unique_id(buf) != 0:
                raise RuntimeError("mot_comm_unique_id failed (librccl not loadable?)")
        raw = bytes(buf)
        if self.world > 1:
            raw = (exchange or self._torch_exchange)(raw if self.rank == 0 else None)
        self.h = C.c_void_p()
        self.lib.mot_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        ctx._chk(self.lib.mot_comm_create(ctx.h, self.world, self.rank, raw, C.byref(self.h)))

    @staticmethod
    def _torch_exchange(raw):
        import torch.distributed as dist
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def gather_tables(self, d_rows, d_counts, nstreams, d_rows_all, rows_cap):
        """device pointers in, device pointer out (see mot_comm_gather_tables); returns (counts [world, S], rows per rank [world])"""
        import ctypes as C
        counts = np.zeros((self.world, nstreams), np.int32)
        per_rank = np.zeros(self.world, np.int32)
        self.lib.mot_comm_gather_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_comm_gather_tables(self.h, C.c_void_p(d_rows), C.c_void_p(d_counts), int(nstreams), C.c_void_p(d_rows_all),
                                                      int(rows_cap), counts.ctypes.data_as(C.c_void_p), per_rank.ctypes.data_as(C.c_void_p)))
        return counts, per_rank

    def close(self):
        if self.h:
            self.lib.mot_comm_destroy(self.h)
            self.h = None
