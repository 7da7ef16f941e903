"""The native track-table gather (mot_comm_gather_tables: RCCL called from the library, exact sizes) with TWO ranks, one per GPU.
Skipped unless the box has at least two GPUs (the 1-GPU boxes of the build loop run it as a skip; the 8-GPU node runs it)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_main(rank, world, idfile, outdir):
    sys.path.insert(0, ROOT)
    import time

    import torch

    from motcpp_amd import _lib as L
    from motcpp_amd import dist as mdist
    from motcpp_amd.synth import SynthStream
    torch.cuda.set_device(rank)
    S, maxd = 3, 96
    dev = L.DeviceByteTrack(S, 512, maxd, device=rank)
    streams = [SynthStream(120, 70, 900 + rank * S + s) for s in range(S)]
    rows = L.pinned_array(dev.ctx, (S * maxd * 2, 8), np.float32)
    cnt = L.pinned_array(dev.ctx, (S,), np.int32)
    ddets = torch.zeros((S, 6, maxd), dtype=torch.float32, device=f"cuda:{rank}")

    def exchange(raw):  # the 128-byte RCCL id through a file (rank 0 writes, the others wait for it)
        if raw is not None:
            with open(idfile + ".tmp", "wb") as f:
                f.write(raw)
            os.replace(idfile + ".tmp", idfile)
            return raw
        for _ in range(600):
            if os.path.exists(idfile):
                return open(idfile, "rb").read()
            time.sleep(0.1)
        raise RuntimeError("no RCCL id from rank 0")
    comm = mdist.NativeComm(dev.ctx, world=world, rank=rank, exchange=exchange)
    all_rows = torch.zeros((world * S * maxd * 2, 8), dtype=torch.float32, device=f"cuda:{rank}")
    for f in range(12):
        counts = np.zeros(S, np.int32)
        soa = np.zeros((S, 6, maxd), np.float32)
        for s, st in enumerate(streams):
            d, _ = st.next_frame()
            counts[s] = len(d)
            soa[s, :, :len(d)] = d.T
        ddets.copy_(torch.from_numpy(soa))
        torch.cuda.synchronize()
        total = dev.step_packed(ddets.data_ptr(), counts, rows, cnt)
    r_ptr, _o_ptr, c_ptr = dev.device_output()
    counts_all, per_rank = comm.gather_tables(r_ptr, c_ptr, S, all_rows.data_ptr(), all_rows.shape[0])
    dev.ctx._chk(dev.ctx.lib.mot_ctx_sync(dev.ctx.h))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), own_rows=rows[:total].copy(), own_cnt=np.array(cnt), counts_all=counts_all, per_rank=per_rank,
             gathered=all_rows[:int(per_rank.sum())].cpu().numpy())
    comm.close()
    dev.close()


@pytest.mark.gpu
def test_native_gather_with_two_ranks(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_rank_main, args=(2, str(tmp_path / "rccl_id"), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    want = np.concatenate([r[0]["own_rows"], r[1]["own_rows"]])
    for k in range(2):  # every rank holds both ranks' tables, rank 0's rows first, exact sizes
        assert r[k]["per_rank"].tolist() == [len(r[0]["own_rows"]), len(r[1]["own_rows"])]
        assert np.array_equal(r[k]["counts_all"][0], r[0]["own_cnt"]) and np.array_equal(r[k]["counts_all"][1], r[1]["own_cnt"])
        assert np.array_equal(r[k]["gathered"], want)
