#!/usr/bin/env python3
"""Benchmark of the association hot path behind tracker.update() (BASELINE.json metric: frames/s at
N_tracks x M_dets with assignment indices identical to the reference path).

A "step" = one frame for every one of the S independent streams a rank owns, stepped in lockstep through
motcpp::StreamBatch (one kernel launch per kernel family per stage, whatever S is). value = frames processed by
all ranks / wall time of the timed region. The detection payload is resident in HBM before the timed region starts
(the host keeps its own copy for the lifecycle decisions); only per-stage index lists / task descriptors and the
result tables cross PCIe inside it. Streams are sharded over GPUs with no data-path collective; the final track
tables are gathered over RCCL every --gather-every steps (weak scaling: S streams per GPU).

Prints ONE JSON line on rank 0 (contract in the task brief) with the extra objects `roofline` (dominant kernel by
summed HIP-event time inside the timed region) and `cpu_baseline` (the CPU oracle = restatement of the reference,
timed on one host core on a bounded sample of stream 0).
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32 MFMA = fp32 vector peak
DENSE_COS = os.environ.get("MOT_BOT_DENSE_COSINE", "") == "1"  # (A/B: BoT-SORT's appearance term as the whole matrix on the MFMA, bot_device.hip)

WORKLOADS = {
    # name: (tracker, P persistent objects, M dets/frame, emb_dim, description)
    "C2": ("bytetrack", 256, 128, 0, "ByteTrack, synthetic 256 tracks x 128 dets/frame, IoU-only cost (BASELINE configs[1])"),
    "NS": ("bytetrack", 1000, 500, 0, "ByteTrack, synthetic 1000 tracks x 500 dets/frame (north-star shape)"),
    "C5": ("bytetrack", 1000, 512, 0, "ByteTrack, 1000 x 512 per stream (BASELINE configs[4] per-GPU shape)"),
    "C3": ("botsort", 1024, 512, 256, "BoT-SORT, 1024 x 512 with 256-d embeddings (BASELINE configs[2])"),
    "C4": ("ocsort", 4096, 2048, 0, "OC-SORT, 4096 x 2048 (BASELINE configs[3])"),
    "SORT": ("sort", 256, 128, 0, "SORT, 256 x 128"),
}


def cpu_budget():
    """CPUs this process may use on average: the cgroup quota if there is one, else the online CPUs."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def world_local():
    """ranks sharing this node (torchrun sets LOCAL_WORLD_SIZE)"""
    return int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: re-executes this command line under torch.distributed.run, one
    process per GPU of this node (rank r drives GPU r through LOCAL_RANK), rendezvous on 127.0.0.1 and a free port. The ranks
    read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment exactly as when the driver starts them itself."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def gather_rank_rates(dist, torch, rate, streams, device):
    """every rank's own rate (frames of ITS streams / ITS wall time of the timed region) on every rank: one all-reduce of a vector with
    one slot per rank, over the job's process group (RCCL on the GPU boxes) — what the line reports as `per_rank`"""
    world, rank = dist.get_world_size(), dist.get_rank()
    tr = torch.zeros(world, dtype=torch.float64, device=device)
    tr[rank] = rate
    dist.all_reduce(tr, op=dist.ReduceOp.SUM)
    return {"frames/s": [float(x) for x in tr.tolist()], "ranks": world, "backend": dist.get_backend(), "streams_per_rank": streams}


def launch_probe(real_stdout):
    """MOT_BENCH_LAUNCH_PROBE=1 (tests/test_bench_contract.py): the ranks only prove that they exist — a gloo rendezvous, every rank's
    pid and LOCAL_RANK gathered, rank 0 prints them as the JSON line. No GPU is touched: this is how the launcher is tested here."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    me = {"rank": dist.get_rank(), "local_rank": int(os.environ.get("LOCAL_RANK", "-1")), "pid": os.getpid()}
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, me)
    import torch
    per_rank = gather_rank_rates(dist, torch, 1000.0 * (dist.get_rank() + 1), 1, "cpu")  # (the same exchange the real run makes, on gloo)
    if dist.get_rank() == 0:
        os.write(real_stdout, (json.dumps({"probe": True, "world": dist.get_world_size(), "ranks": box, "per_rank": per_rank}) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


KF_BT_UPDATE_B = 2 * (32.0 + 64.0) + 16.0 + 8.0 + 2.0  # ByteTrack's device lifecycle: mean + covariance blocks in and out, measurement, indices, flags
KF_BT_BIRTH_B = 32.0 + 64.0 + 16.0 + 1.0
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
try:
    from kernel_sources_hash import kernel_sources_hash
    KSRC = kernel_sources_hash()
except Exception:  # noqa: BLE001
    KSRC = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 60; C4: 7 - a step there is 256 assignment problems of 4096 x 2048)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 40; C4: 1)")
    ap.add_argument("--workload", default="NS", choices=sorted(WORKLOADS),
                    help="NS = the north-star shape the >= 50 k frames/s bar is set on (ByteTrack, 1000 tracks x 500 detections)")
    ap.add_argument("--settle", type=int, default=None,
                    help="untimed frames stepped before the warm-up so that the track pools are in steady state whatever --warmup is "
                         "(a stream starts by giving birth to its whole population at once)")
    ap.add_argument("--host-input-steps", type=int, default=8,
                    help="after the timed region: this many more steps with the detections uploaded from page-locked host memory inside "
                         "the step (reported as host_input; 0 = skip). Device-lifecycle workloads only")
    ap.add_argument("--isolated-steps", type=int, default=None,
                    help="after the timed region: this many steps with the sub-batches one after the other, for per-kernel times "
                         "without time-sharing (reported as kernels_isolated; default 4, C4: 0)")
    ap.add_argument("--no-in-flight", dest="in_flight", action="store_false",
                    help="ByteTrack and BoT-SORT on the device keep two frames in flight per sub-batch from ONE host thread (mot_bt_enqueue_packed / "
                         "mot_bt_collect_packed: the result copy of frame f overlaps the kernels of frame f + 1; measured 1.6 M against "
                         "1.13 M frames/s at the north-star shape); this flag goes back to one driver thread per sub-batch waiting for each frame")
    ap.add_argument("--streams", type=int, default=0, help="independent streams per GPU (0: workload default)")
    ap.add_argument("--threads", type=int, default=0, help="host worker threads for the per-stream lifecycle, shared by the sub-batches (0: min(64, cores))")
    ap.add_argument("--gather-every", type=int, default=8)
    ap.add_argument("--gather", default=None, choices=["torch", "native"],
                    help="how the ranks' packed track tables are gathered: torch = torch.distributed.all_gather_into_tensor on zero-copy views of "
                         "the library's device buffers (padded to a fixed size); native = mot_comm_gather_tables, RCCL called from the library on "
                         "the sub-batch's stream with exact sizes (also runs with one rank, as a self-test)")
    ap.add_argument("--lifecycle", choices=["auto", "device", "host"], default="auto",
                    help="where the per-stream track bookkeeping runs: device = mot_bt_* (ByteTrack only: four small kernels per frame, "
                         "no host decisions), host = the C++ stage machines; auto = device where it exists")
    ap.add_argument("--pin", type=int, default=1, help="pin each sub-batch's host worker team to its own consecutive CPUs")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="split a rank's streams into this many sub-batches with their own HIP stream, stepped concurrently so one's host lifecycle overlaps another's kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-outputs-resident", action="store_true", help="skip the tables-stay-in-HBM leg behind the timed region (profiling runs whose last launches "
                    "should be the timed ones: tools/last_launches_avg.py)")
    ap.add_argument("--parity-streams", type=int, default=None,
                    help="streams of rank 0 (seeded sample over all sub-batches, stream 0 among them) whose outputs are compared with the oracle "
                         "(default 32; C3: 8, C4: 4 - the oracle runs 17 / 0.2 frames/s there)")
    ap.add_argument("--sweep-streams", default=None,
                    help="comma-separated stream counts of the S-sweep after the timed region (default 1,64,1024 for the device lifecycles; '' = off)")
    ap.add_argument("--long-run-steps", type=int, default=None,
                    help="steps of the long_run leg after the timed region (default 300; C4: 0): the resident frames played back and forth")
    ap.add_argument("--trace-steps", action="store_true", help="add the wall time of every timed step to the line (step_ms)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args.gpus)  # does not return: this process becomes the launcher of one rank per GPU
    # stdout carries exactly one JSON line: libraries that print there (RCCL's version banner at communicator creation) are
    # sent to stderr for the whole run, the line itself goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if os.environ.get("MOT_BENCH_LAUNCH_PROBE") == "1" and "RANK" in os.environ:
        launch_probe(real_stdout)
        return
    if args.gather is None:  # several ranks: the library's own RCCL gather (exact sizes, on the sub-batch's stream)
        args.gather = "native" if world > 1 else "torch"

    import torch
    import torch.distributed as dist
    from motcpp_amd import _lib as L
    from motcpp_amd import dist as mdist
    from motcpp_amd.synth import SynthStream

    if not (os.path.exists(L.HIP_LIB) and os.path.exists(L.HOST_LIB)):
        L.build()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible (--gpus {args.gpus})")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    tracker, P, M, D, desc = WORKLOADS[args.workload]
    S = args.streams or {"C2": 12288, "SORT": 12288, "NS": 18432, "C5": 18432, "C3": 1536, "C4": 768}[args.workload]
    # host workers block between phases, so about twice as many workers as the box's CPU quota pay off (the bursts of
    # lifecycle work get shorter and the workers sleep through the GPU waits); far more than that and the cgroup
    # throttles the whole process (measured on the 16-CPU-quota GPU boxes: 32 workers 456k frames/s, 64 workers 268k)
    threads = args.threads or max(2, min(os.cpu_count() or 1, 64, 2 * cpu_budget() // max(1, world_local())))
    heavy = args.workload == "C4"  # seconds per step: keep the default run within minutes
    K = args.steps if args.steps is not None else (7 if heavy else 60)
    W = args.warmup if args.warmup is not None else (1 if heavy else 40)
    if args.settle is None:
        args.settle = 12 if heavy else 30  # (C4: 12 + 1 + 7 = the 20 frames per stream the parity sample covers)
    if args.isolated_steps is None:
        args.isolated_steps = 0 if heavy else 4
    on_device_wl = tracker in ("bytetrack", "sort") and args.lifecycle in ("auto", "device")  # (host-input leg: detections only)
    Z = max(0, args.settle)
    H = max(0, args.host_input_steps) if on_device_wl else 0
    W0 = W            # the warm-up the caller asked for (reported); the settling frames are stepped before it
    W = W + Z
    ISO = max(0, args.isolated_steps)
    F = K + W + H + ISO

    # ---- synthetic streams (seed 1234 + global stream id), generated before anything is timed ----
    # (round 5: the streams are generated by a pool of fresh interpreter processes — 18 432 streams x 67 frames took minutes on one core; every
    # stream has its own seeded generator, so the arrays are the same whatever the number of workers)
    from motcpp_amd.synth import generate_streams
    gen_workers = max(1, min(16, cpu_budget() // max(1, world_local())))
    host, embs = generate_streams(P, M, D, [mdist.stream_seed(g) for g in mdist.stream_ids(rank, S)], F, gen_workers)
    # detection payload resident in HBM as SoA [F, S, 6, M] before the timed region
    dev_dets = torch.from_numpy(np.ascontiguousarray(host.transpose(0, 1, 3, 2))).cuda(local)
    dev_embs = torch.from_numpy(embs).cuda(local) if D else None  # [F, S, M, D] row-major, resident like the detections
    torch.cuda.synchronize()
    frame_bytes = S * 6 * M * 4

    if args.pipeline <= 0:
        args.pipeline = {"C2": 3, "SORT": 3, "NS": 3, "C5": 3, "C3": 3, "C4": 3}.get(args.workload, 2)  # measured on MI355X
    PIPE = max(1, min(args.pipeline, S))
    bounds = [S * p // PIPE for p in range(PIPE + 1)]
    on_device = tracker in ("bytetrack", "sort", "botsort", "ocsort") and args.lifecycle in ("auto", "device")
    if args.lifecycle == "device" and not on_device:
        raise SystemExit("--lifecycle device exists for the ByteTrack, SORT, BoT-SORT and OC-SORT workloads only")
    if on_device:
        cap_tracks = (2 * P + 63) // 64 * 64  # tracked + lost never get near twice the object count (else mot_bt_step reports it)
        if tracker == "botsort":
            batches = [L.DeviceBotSort(bounds[p + 1] - bounds[p], cap_tracks, M, D, device=local) for p in range(PIPE)]
        elif tracker == "ocsort":
            batches = [L.DeviceOCSort(bounds[p + 1] - bounds[p], cap_tracks, M, device=local) for p in range(PIPE)]
        else:
            Dev = L.DeviceByteTrack if tracker == "bytetrack" else L.DeviceSort
            batches = [Dev(bounds[p + 1] - bounds[p], cap_tracks, M, device=local) for p in range(PIPE)]
        full_counts = [np.full(bounds[p + 1] - bounds[p], M, np.int32) for p in range(PIPE)]
    else:
        batches = [L.Batch(tracker, bounds[p + 1] - bounds[p], device=local, threads=max(1, threads // PIPE), record_laps=False,
                           private_device=PIPE > 1) for p in range(PIPE)]
    cap = max(2 * M, 64)
    gathered = None
    # ByteTrack on the device: packed output (mot_bt_step_packed) — the emitted rows of a sub-batch back to back, so that only
    # rows that exist cross PCIe (a padded [S, 2M, 8] table is 2-4x the bytes) and no stream has a row limit
    packed = on_device  # (all four device lifecycles emit packed tables)
    rows_cap = [int((bounds[p + 1] - bounds[p]) * (M if tracker == "bytetrack" else max(M, P)) * 1.25) + 64 for p in range(PIPE)]
    if packed:
        rows_p = [torch.zeros((rows_cap[p], 8), dtype=torch.float32).pin_memory().numpy() for p in range(PIPE)]
        tot_p = [0] * PIPE
        out_all = None
        cnt_all = torch.zeros((S,), dtype=torch.int32).pin_memory().numpy()
    elif on_device:  # page-locked, so that the one result copy of a frame runs at PCIe speed
        out_all = torch.zeros((S, cap, 8), dtype=torch.float32).pin_memory().numpy()
        cnt_all = torch.zeros((S,), dtype=torch.int32).pin_memory().numpy()
    else:
        out_all = np.zeros((S, cap, 8), np.float32)
        cnt_all = np.zeros(S, np.int32)
    from concurrent.futures import ThreadPoolExecutor
    # one dedicated driver thread per sub-batch: its worker team (and the CPUs that team is pinned to) never changes
    pools = [ThreadPoolExecutor(1) for _ in range(PIPE)] if PIPE > 1 else None
    tpb = max(1, threads // PIPE)
    if args.pin and not on_device:
        for p in range(PIPE):
            first = local * threads + p * tpb
            if pools is None:
                batches[p].pin_threads(first)
            else:
                pools[p].submit(batches[p].pin_threads, first).result()

    def sub_step(p, f):
        s0, s1 = bounds[p], bounds[p + 1]
        if packed and tracker == "botsort":  # detections and raw embeddings resident: [S][6][M] and [S][M][D]
            tot_p[p] = batches[p].step_packed(dev_dets.data_ptr() + (f * S + s0) * 6 * M * 4, full_counts[p], rows_p[p], cnt_all[s0:s1],
                                              embs_ptr=(dev_embs.data_ptr() + (f * S + s0) * M * D * 4) if D else None)
            return
        if packed:
            tot_p[p] = batches[p].step_packed(dev_dets.data_ptr() + (f * S + s0) * 6 * M * 4, full_counts[p], rows_p[p], cnt_all[s0:s1])
            return
        if on_device:  # tables land directly in this sub-batch's slice of the rank's output
            batches[p].step(resident_ptr=dev_dets.data_ptr() + (f * S + s0) * 6 * M * 4, counts=full_counts[p],
                            out=out_all[s0:s1], out_counts=cnt_all[s0:s1])
            return
        else:
            o, c = batches[p].step(host[f, s0:s1], embs=None, cap=cap,
                                   resident_ptr=dev_dets.data_ptr() + (f * S + s0) * 6 * M * 4,
                                   resident_embs=(dev_embs.data_ptr() + (f * S + s0) * M * D * 4, D) if D else None)
        out_all[s0:s1] = o[:, :cap]
        cnt_all[s0:s1] = c

    def step(f):
        if pools is None:
            sub_step(0, f)
        else:
            for fut in [pools[p].submit(sub_step, p, f) for p in range(PIPE)]:
                fut.result()
        return out_all, cnt_all

    def counters():
        tot = {}
        if on_device:  # no flushes: a frame is a fixed sequence of 14 launches + 1 result copy per sub-batch
            return {"flushes": 0, "launches": 0, "ms_begin": 0.0, "ms_flush": 0.0, "ms_advance": 0.0, "ms_sync_wait": 0.0}
        for b in batches:
            for k, v in b.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    comms, native_bufs = None, None

    def gather(out, cnt):
        # final track tables of this rank's streams -> every rank (RCCL all_gather over xGMI)
        nonlocal gathered, comms, native_bufs
        if packed and args.gather == "native":  # RCCL from the library, one communicator per sub-batch (its context's stream)
            res = []
            for p in range(PIPE):
                r_ptr, _o_ptr, c_ptr = batches[p].device_output()
                res.append(comms[p].gather_tables(r_ptr, c_ptr, bounds[p + 1] - bounds[p], native_bufs[p].data_ptr(), world * rows_cap[p]))
            for p in range(PIPE):
                batches[p].ctx._chk(batches[p].lib.mot_ctx_sync(batches[p].ctx.h))
            gathered = (native_bufs, res)
            return
        if packed:  # straight from the device-resident packed tables of the last step: no copy through the host
            dv = torch.device("cuda", local)
            rl, cl = [], []
            for p in range(PIPE):
                r_ptr, _o_ptr, c_ptr = batches[p].device_output()
                rl.append(mdist.device_view(r_ptr, (tot_p[p], 8), torch.float32, dv) if tot_p[p] else torch.zeros((0, 8), device=dv))
                cl.append(mdist.device_view(c_ptr, (bounds[p + 1] - bounds[p],), torch.int32, dv))
            gathered = mdist.gather_packed(rl, cl, sum(rows_cap))
            return
        gathered = mdist.gather_tables(out[:, :cap], cnt.astype(np.int32), device=torch.device("cuda", local))

    # parity sample: a seeded choice of this rank's streams, stream 0 and the first stream of every sub-batch among them
    n_par = args.parity_streams if args.parity_streams is not None else {"C3": 8, "C4": 4}.get(args.workload, 32)
    n_par = max(1, min(n_par, S))
    forced = sorted({0} | {bounds[p] for p in range(PIPE)})[:n_par]
    rest = [int(x) for x in np.random.default_rng(20240903).permutation(S) if int(x) not in forced]
    parity_ids = sorted(forced + rest[:n_par - len(forced)])
    parity_sub = [max(q for q in range(PIPE) if bounds[q] <= sid) for sid in parity_ids]

    def sample_rows(out, cnt):
        """the emitted rows of the sampled streams after the frame just collected"""
        got = {}
        if packed:
            offs = [np.concatenate(([0], np.cumsum(cnt[bounds[p]:bounds[p + 1]]))) for p in range(PIPE)]
        for sid, p in zip(parity_ids, parity_sub):
            if packed:
                off = int(offs[p][sid - bounds[p]])
                got[sid] = rows_p[p][off:off + cnt[sid]].copy()
            else:
                got[sid] = out[sid, :cnt[sid]].copy()
        return got

    in_flight = args.in_flight and packed and not heavy  # (round 3: mot_sort_* and mot_oc_* have enqueue / collect as well; C4 is one long kernel per frame: nothing to overlap)
    # C4 (round 5): a frame of a sub-batch is ONE launch of the exact assignment kernel that lasts as long as its slowest problem (one problem per CU,
    # 0.25 - 0.36 s each): stepped in lockstep, every CU that finished early idled until the slowest problem of ALL sub-batches was done. Each sub-batch's
    # driver thread now steps its own frames without waiting for the others (still one frame at a time per stream, K frames each inside the timed
    # region), with more streams than CUs queued: a CU that finishes a problem takes the next one of whichever sub-batch has one waiting.
    free_running = heavy and packed and pools is not None and world == 1

    def sample_rows_sub(p):
        got = {}
        offs = np.concatenate(([0], np.cumsum(cnt_all[bounds[p]:bounds[p + 1]])))
        for sid, q in zip(parity_ids, parity_sub):
            if q == p:
                off = int(offs[sid - bounds[p]])
                got[sid] = rows_p[p][off:off + cnt_all[sid]].copy()
        return got

    def run_free(f0, n, keep_limit, marks):
        """frames f0 .. f0+n-1 of every sub-batch, each on its own driver thread at its own pace; returns when all are done"""
        kept_p = [[] for _ in range(PIPE)]
        marks_p = [[] for _ in range(PIPE)]

        def drive(p):
            for k in range(n):
                sub_step(p, f0 + k)
                marks_p[p].append(time.perf_counter())
                if rank == 0 and k < keep_limit:
                    kept_p[p].append(sample_rows_sub(p))
        for fut in [pools[p].submit(drive, p) for p in range(PIPE)]:
            fut.result()
        for k in range(min(n, keep_limit) if rank == 0 else 0):
            merged = {}
            for p in range(PIPE):
                merged.update(kept_p[p][k])
            kept.append(merged)
        if marks is not None:
            for k in range(n):  # a step is complete when its frame is done in every sub-batch
                marks.append(max(marks_p[p][k] for p in range(PIPE)))

    def run_pipelined(f0, n, keep_limit):
        """frames f0 .. f0+n-1 with two frames in flight per sub-batch, one host thread; returns when the last one is collected"""
        def enq(f):
            for p in range(PIPE):
                dp = dev_dets.data_ptr() + (f * S + bounds[p]) * 6 * M * 4
                if tracker == "botsort":
                    batches[p].enqueue_packed(dp, full_counts[p], rows_cap[p], embs_ptr=(dev_embs.data_ptr() + (f * S + bounds[p]) * M * D * 4) if D else None)
                else:
                    batches[p].enqueue_packed(dp, full_counts[p], rows_cap[p])

        def col(k):
            for p in range(PIPE):
                tot_p[p] = batches[p].collect_packed(rows_p[p], cnt_all[bounds[p]:bounds[p + 1]])
            if f0 >= W:
                step_marks.append(time.perf_counter())
            if rank == 0 and k < keep_limit:
                kept.append(sample_rows(None, cnt_all))
            if (world > 1 or args.gather == "native") and f0 >= W and ((k + 1) % args.gather_every == 0 or k == n - 1):
                gather(None, cnt_all)
        enq(f0)
        for k in range(1, n):
            enq(f0 + k)
            col(k - 1)
        col(n - 1)

    if packed and args.gather == "native":  # communicators are created outside the timed region (RCCL initialisation takes seconds)
        comms = [mdist.NativeComm(batches[p].ctx, world=world, rank=rank) for p in range(PIPE)]
        native_bufs = [torch.empty((world * rows_cap[p], 8), dtype=torch.float32, device=f"cuda:{local}") for p in range(PIPE)]
    kept = []  # per kept frame: the outputs of rank 0's sampled streams (parity check against the oracle)
    step_marks = []  # wall clock after every timed step (collected frame)
    if in_flight:
        run_pipelined(0, W, 40)
        out, cnt = None, cnt_all
    elif free_running:
        run_free(0, W, 40, None)
        out, cnt = None, cnt_all
    else:
        for f in range(W):
            out, cnt = step(f)
            if rank == 0 and f < 40:
                kept.append(sample_rows(out, cnt))
    n_kept_warm = len(kept)
    if world > 1:
        gather(out, cnt)
        dist.barrier()
    torch.cuda.synchronize()
    for b in batches:
        b.profile(True)
    diag_ctx = L.Context(local)
    diag_ctx.lap_fast_stats(reset=True)
    gc.collect()
    gc.disable()  # (a collection of the interpreter's older generations in the middle of the timed region is tens of ms)
    c0 = counters()
    t0 = time.perf_counter()
    if in_flight:
        run_pipelined(W, K, 8)
    elif free_running:
        run_free(W, K, 8, step_marks)
    else:
        for k in range(K):
            out, cnt = step(W + k)
            step_marks.append(time.perf_counter())
            if world > 1 and ((k + 1) % args.gather_every == 0 or k == K - 1):
                gather(out, cnt)
            if rank == 0 and k < 8:
                kept.append(sample_rows(out, cnt))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    gc.enable()
    achieved_dims = None
    if on_device and tracker == "bytetrack":
        dims = np.sum([b.profile_dims() for b in batches], axis=0)
        pc = [sum(b.profile_stats()[k] for b in batches) for k in ("lap1_problems", "lap23_problems")]
        achieved_dims = {"first_association": {"problems": int(pc[0]), "mean_tracks_N": dims[0] / max(pc[0], 1), "mean_dets_M": dims[1] / max(pc[0], 1)},
                    "second_and_unconfirmed": {"problems": int(pc[1]), "mean_tracks_N": dims[2] / max(pc[1], 1), "mean_dets_M": dims[3] / max(pc[1], 1)},
                    "note": "rows x columns of the assignment problems actually queued in the timed region (pool of tracked + lost tracks x "
                            "high-score detections; then remaining tracked x low-score detections and unconfirmed x remaining detections)"}
    def collect_families():
        """per kernel family: summed HIP-event ms, launches, tasks, algorithmic bytes / flops since profile(True); stops profiling"""
        acc = {}
        for b in batches:
            ps = b.profile_stats()
            if on_device and tracker == "ocsort":
                ps = {"lap": {"ms": ps["lap_ms"], "launches": ps["frames"], "tasks": ps["lap_problems"], "bytes": 4.0 * ps["lap_nm"], "flops": 0.0},
                      "ocsort_cost": {"ms": ps["cost_ms"], "launches": ps["frames"], "tasks": ps["lap_problems"], "bytes": 8.0 * ps["lap_nm"], "flops": 0.0},
                      "frame_all_kernels": {"ms": ps["frame_ms"], "launches": ps["frames"], "tasks": (bounds[1] - bounds[0]) * ps["frames"],
                                            "bytes": 0.0, "flops": 0.0}}
            elif on_device and tracker == "botsort":
                ps = {"lap": {"ms": ps["lap_ms"], "launches": 2 * ps["frames"], "tasks": ps["lap_problems"], "bytes": 24.0 * ps["lap_nm"], "flops": 0.0},
                      # the first association's appearance term. Round 5: embed_gated_kernel — the cosine distance of the pairs that pass the
                      # proximity test only (about one per track), so there is no n x m contraction to count: bytes = the boxes in (20 B per row
                      # and column) + two feature rows in and one distance out per pair that passes, taken as one pair per detection
                      # (MOT_BOT_DENSE_COSINE=1 runs the fp32 MFMA matrix kernel instead: its flops are 2 n m D)
                      "cosine": {"ms": ps["cos_ms"], "launches": ps["frames"], "tasks": ps["frames"] * (bounds[1] - bounds[0]),
                                 "bytes": (4.0 * (ps["cos_nm"] + ps["frames"] * (bounds[1] - bounds[0]) * (P + M) * D) if DENSE_COS else
                                           ps["frames"] * (bounds[1] - bounds[0]) * (20.0 * (P + M) + min(P, M) * (8.0 * D + 4.0))),
                                 "flops": (2.0 * ps["cos_nm"] * D) if DENSE_COS else 0.0},
                      # feat_kernel: the three launches of a frame (normalise the detections' rows, set the new tracks' features, blend the matched
                      # ones); bytes = 4 D per row read or written (normalise / set: 2 per row, blend: 3), counted on the device
                      "feat": {"ms": ps["feat_ms"], "launches": 3 * ps["frames"], "tasks": 3 * ps["frames"] * (bounds[1] - bounds[0]),  # (a task per stream and launch)
                               "bytes": 4.0 * D * ps["feat_row_moves"], "flops": 0.0},
                      "frame_all_kernels": {"ms": ps["frame_ms"], "launches": ps["frames"], "tasks": (bounds[1] - bounds[0]) * ps["frames"],
                                            "bytes": 0.0, "flops": 0.0}}
            elif on_device:  # the solver's launches are timed on the device stream; "frame" = all 14 launches of a frame
                ps_raw = ps
                ps = {"lap": {"ms": ps["lap1_ms"] + ps["lap23_ms"], "launches": (2 if tracker == "bytetrack" else 1) * ps["frames"], "tasks": ps["lap1_problems"] + ps["lap23_problems"],
                              "bytes": 24.0 * (ps["lap1_nm"] + ps["lap23_nm"]), "flops": 0.0},
                      "frame_all_kernels": {"ms": ps["frame_ms"], "launches": ps["frames"], "tasks": (bounds[1] - bounds[0]) * ps["frames"],
                                            "bytes": 0.0, "flops": 0.0}}
                if tracker == "bytetrack":  # the Kalman launches of the same frames (bytes: DESIGN.md's per-item figures)
                    sp = b.profile_lap_sparse()  # ONE kernel: the first association's sparse solver, events around that launch only
                    ps["lap1_sparse"] = {"ms": sp["ms"], "launches": sp["launches"], "tasks": ps_raw["lap1_problems"],
                                         "bytes": 24.0 * ps_raw["lap1_nm"], "flops": 0.0}
                    kf = b.profile_kalman()
                    fr = ps["frame_all_kernels"]["launches"]
                    # (round 5: the predicted boxes of the pool are computed inside bt_begin — no launch of their own any more; the item count stays for the byte model)
                    ps["kf_predict_boxes"] = {"ms": 0.0, "launches": fr, "tasks": kf["predict_boxes_items"], "bytes": 52.0 * kf["predict_boxes_items"], "flops": 0.0}
                    ps["kf_initiate"] = {"ms": 0.0,  # (round 5: written by bt_after_second, no launch of its own)
                                         "launches": fr, "tasks": kf["initiate_items"], "bytes": KF_BT_BIRTH_B * kf["initiate_items"], "flops": 0.0}
                    # (round 6: block-form covariances — 32 B of mean + 64 B of blocks in and out, the measurement, two indices, two flag bytes)
                    ps["kf_update"] = {"ms": kf["update_ms"], "launches": fr, "tasks": kf["update_items"], "bytes": KF_BT_UPDATE_B * kf["update_items"], "flops": 0.0}
            for k, v in ps.items():
                a = acc.setdefault(k, {"ms": 0.0, "launches": 0, "tasks": 0, "bytes": 0.0, "flops": 0.0})
                for kk in a:
                    a[kk] += v[kk]
            b.profile(False)
        return acc

    stats = collect_families()
    c1 = counters()
    fast_stats = diag_ctx.lap_fast_stats()
    elapsed = t1 - t0
    # ---- the same steps with the detections coming from (page-locked) host memory inside the step ----
    host_input = None
    if H > 0 and on_device:
        pinned = torch.from_numpy(np.ascontiguousarray(host[W + K:W + K + H].transpose(0, 1, 3, 2))).pin_memory()  # [H, S, 6, M]
        stage = [torch.empty((bounds[p + 1] - bounds[p], 6, M), dtype=torch.float32, device=f"cuda:{local}") for p in range(PIPE)]

        def sub_step_host(p, h):
            s0, s1 = bounds[p], bounds[p + 1]
            b = batches[p]
            src = pinned[h, s0:s1]
            b.ctx._chk(b.lib.mot_memcpy_h2d(b.ctx.h, C.c_void_p(stage[p].data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(src.numel() * 4)))
            if packed:
                tot_p[p] = b.step_packed(stage[p].data_ptr(), full_counts[p], rows_p[p], cnt_all[s0:s1])
            else:
                b.step(resident_ptr=stage[p].data_ptr(), counts=full_counts[p], out=out_all[s0:s1], out_counts=cnt_all[s0:s1])

        torch.cuda.synchronize()
        if in_flight:  # the same two-frames-in-flight pipeline, each frame preceded by its H2D copy on the sub-batch's stream
            stage2 = [[stage[p], torch.empty_like(stage[p])] for p in range(PIPE)]

            def enq_h(h):
                for p in range(PIPE):
                    b = batches[p]
                    src = pinned[h, bounds[p]:bounds[p + 1]]
                    dst = stage2[p][h & 1]
                    b.ctx._chk(b.lib.mot_memcpy_h2d(b.ctx.h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(src.numel() * 4)))
                    b.enqueue_packed(dst.data_ptr(), full_counts[p], rows_cap[p])

            def col_h():
                for p in range(PIPE):
                    tot_p[p] = batches[p].collect_packed(rows_p[p], cnt_all[bounds[p]:bounds[p + 1]])
            th0 = time.perf_counter()
            enq_h(0)
            for h in range(1, H):
                enq_h(h)
                col_h()
            col_h()
        else:
            th0 = time.perf_counter()
            for h in range(H):
                if pools is None:
                    sub_step_host(0, h)
                else:
                    for fut in [pools[p].submit(sub_step_host, p, h) for p in range(PIPE)]:
                        fut.result()
        torch.cuda.synchronize()
        th1 = time.perf_counter()
        host_input = {"value": S * H / (th1 - th0), "unit": "frames/s (this rank)", "steps": H, "ms_per_step": (th1 - th0) / H * 1e3,
                      "h2d_bytes_per_step": S * 6 * M * 4,
                      "note": "same tracker state, the frames after the timed region; detections copied from page-locked host memory on the "
                              "sub-batch's stream inside each step (PCIe-inclusive rate; never the headline value)"}
    # ---- the same kernels without the sub-batches time-sharing the GPU: a few more steps, one sub-batch at a time ----
    # (inside the timed region three sub-batches run on their own HIP streams, so a kernel's start-to-end time includes the other
    # sub-batches' kernels; these numbers are what a launch costs when it has the GPU to itself)
    isolated = None
    if ISO > 0:
        for b in batches:
            b.profile(True)
        torch.cuda.synchronize()
        for k in range(ISO):
            for p in range(PIPE):
                sub_step(p, W + K + H + k)
        torch.cuda.synchronize()
        iso = collect_families()
        isolated = {}
        for k, v in iso.items():
            if not v["launches"] or v["ms"] <= 0:
                continue
            e = {"avg_launch_ms": round(v["ms"] / v["launches"], 4)}
            if v["bytes"] > 0:
                e["GB/s"] = round(v["bytes"] / v["ms"] / 1e6, 1)
                e["hbm_frac"] = round(v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS, 4)
            if v["flops"] > 0:
                e["TFLOP/s"] = round(v["flops"] / v["ms"] / 1e9, 2)
                e["mfma_f32_frac"] = round(v["flops"] / v["ms"] / 1e9 / MFMA_F32_PEAK_TFLOPS, 4)
            isolated[k] = e
        isolated["note"] = f"{ISO} steps after the timed region with the sub-batches stepped one after the other (no time-sharing)"
    # ---- long_run: many more steps on the same trackers (does the rate depend on the age of the run?) ----
    long_run = None
    lr_state = None
    LR = args.long_run_steps if args.long_run_steps is not None else (0 if heavy else 300)
    if LR > 0 and on_device and world == 1 and F - Z >= 8:
        cur, lo_f, hi_f = W + K + H + ISO - 1, Z, F - 1
        seq, f, step_dir = [], cur, -1
        while len(seq) < LR:  # the resident frames after the settling ones, played back and forth from where the run stands
            if not lo_f <= f + step_dir <= hi_f:
                step_dir = -step_dir
            f += step_dir
            seq.append(f)
        lr_state = (f, step_dir)  # (outputs_resident goes on from here)

        def enq_lr(f):
            for p in range(PIPE):
                dp = dev_dets.data_ptr() + (f * S + bounds[p]) * 6 * M * 4
                if tracker == "botsort":
                    batches[p].enqueue_packed(dp, full_counts[p], rows_cap[p], embs_ptr=(dev_embs.data_ptr() + (f * S + bounds[p]) * M * D * 4) if D else None)
                else:
                    batches[p].enqueue_packed(dp, full_counts[p], rows_cap[p])

        def col_lr():
            for p in range(PIPE):
                tot_p[p] = batches[p].collect_packed(rows_p[p], cnt_all[bounds[p]:bounds[p + 1]])
        marks = []
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        tl0 = time.perf_counter()
        if in_flight:
            enq_lr(seq[0])
            for f in seq[1:]:
                enq_lr(f)
                col_lr()
                marks.append(time.perf_counter())
            col_lr()
            marks.append(time.perf_counter())
        else:
            for f in seq:
                step(f)
                marks.append(time.perf_counter())
        torch.cuda.synchronize()
        tl1 = time.perf_counter()
        gc.enable()
        sm = np.diff(np.array([tl0] + marks)) * 1e3
        long_run = {"value": S * LR / (tl1 - tl0), "unit": "frames/s (this rank)", "steps": LR, "ms_per_step": (tl1 - tl0) / LR * 1e3,
                    "step_ms_median": float(np.median(sm)), "step_ms_p99": float(np.percentile(sm, 99)), "step_ms_max": float(sm.max()),
                    "steps_over_1.5x_median": int((sm > 1.5 * np.median(sm)).sum()), "vs_value": None,
                    "note": "same trackers, after the timed region: the resident frames behind the settling ones played back and forth "
                            "(object motion stays continuous; velocities flip at the turning points), two frames in flight as in the timed "
                            "region; a step over 1.5x the median = a launch of the exact assignment kernel for a problem the sparse "
                            "solver declined (a non-unique optimum), which the whole sub-batch waits for"}
    # ---- outputs_resident: the same frames with the tables LEFT on the device (a consumer on the GPU / the RCCL gather reads them there) ----
    outputs_resident = None
    if in_flight and on_device and world == 1 and not heavy and F - Z >= 8 and not args.no_outputs_resident:
        nres = min(60, LR if LR > 0 else 60)
        # the same back-and-forth playback as long_run, continued from the frame the trackers stand at. (Round 4 played the frames Z, Z+1, ..., F-1,
        # Z, ... here: every wrap-around teleported all objects - a frame of births, lost tracks and declined assignments - and the leg
        # measured 1.2 M against 2.27 M frames/s for that reason alone, not because the tables stayed on the device.)
        f, step_dir = lr_state if long_run is not None else (W + K + H + ISO - 1, -1)
        fr = []
        while len(fr) < nres:
            if not Z <= f + step_dir <= F - 1:
                step_dir = -step_dir
            f += step_dir
            fr.append(f)

        def enq_r(f):
            for p in range(PIPE):
                dp = dev_dets.data_ptr() + (f * S + bounds[p]) * 6 * M * 4
                if tracker == "botsort":
                    batches[p].enqueue_packed(dp, full_counts[p], rows_cap[p], embs_ptr=(dev_embs.data_ptr() + (f * S + bounds[p]) * M * D * 4) if D else None)
                else:
                    batches[p].enqueue_packed(dp, full_counts[p], rows_cap[p])

        def col_r():
            for p in range(PIPE):
                tot_p[p] = batches[p].collect_packed(None, cnt_all[bounds[p]:bounds[p + 1]])
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        tr0 = time.perf_counter()
        enq_r(fr[0])
        for f in fr[1:]:
            enq_r(f)
            col_r()
        col_r()
        torch.cuda.synchronize()
        tr1 = time.perf_counter()
        gc.enable()
        outputs_resident = {"value": S * nres / (tr1 - tr0), "unit": "frames/s (this rank)", "steps": nres, "ms_per_step": (tr1 - tr0) / nres * 1e3,
                            "note": "same trackers after the long run, mot_*_collect_packed(rows = NULL): per-stream row counts come back, the packed "
                                    "tables stay in HBM (mot_*_device_output) — what a pipeline whose consumer is on the GPU pays; never the headline value"}
    # ---- S-sweep: the same lifecycle with fewer streams, down to ONE stream (the latency of a single tracker.update()) ----
    sweep = None
    sweep_list = [int(x) for x in (args.sweep_streams if args.sweep_streams is not None else ("" if heavy else "1,64,1024")).split(",") if x.strip()]
    if sweep_list and on_device and world == 1:
        sweep = {}
        zs, ks = min(Z, max(0, F - 24)), min(20, F - min(Z, max(0, F - 24)))
        for S2 in sorted(set(x for x in sweep_list if 1 <= x <= S)):
            if tracker == "botsort":
                b2 = L.DeviceBotSort(S2, cap_tracks, M, D, device=local)
            elif tracker == "ocsort":
                b2 = L.DeviceOCSort(S2, cap_tracks, M, device=local)
            else:
                b2 = (L.DeviceByteTrack if tracker == "bytetrack" else L.DeviceSort)(S2, cap_tracks, M, device=local)
            cnt2 = np.full(S2, M, np.int32)
            oc2 = np.zeros(S2, np.int32)
            r2 = torch.zeros((int(S2 * max(M, P) * 1.25) + 64, 8), dtype=torch.float32).pin_memory().numpy()
            out2 = np.zeros((S2, cap, 8), np.float32)

            def one(f):
                dp = dev_dets.data_ptr() + f * S * 6 * M * 4  # streams 0 .. S2-1 of frame f
                if tracker == "botsort":
                    b2.step_packed(dp, cnt2, r2, oc2, embs_ptr=(dev_embs.data_ptr() + f * S * M * D * 4) if D else None)
                elif packed:
                    b2.step_packed(dp, cnt2, r2, oc2)
                else:
                    b2.step(resident_ptr=dp, counts=cnt2, out=out2, out_counts=oc2)
            for f in range(zs):
                one(f)
            torch.cuda.synchronize()
            lat = []
            for f in range(zs, zs + ks):
                ta = time.perf_counter()
                one(f)
                lat.append(time.perf_counter() - ta)
            lat = np.array(lat) * 1e3
            sweep[str(S2)] = {"frames/s": S2 * len(lat) / (lat.sum() * 1e-3), "ms_per_step_median": float(np.median(lat)), "ms_per_step_max": float(lat.max()),
                              "steps": len(lat)}
            b2.close()
        sweep[str(S)] = {"frames/s": S * K / elapsed, "ms_per_step_median": elapsed / K * 1e3, "steps": K,
                         "note": "the timed region (sub-batches and frames in flight as configured)"}
        # The reference's own surface (round 4): T tracker OBJECTS of the public C++ classes (motcpp::trackers::*), one host thread each,
        # every thread calling BaseTracker::update(dets, img) on host detections (motcpp_bench_threads, csrc/host/bench_threads.cpp).
        # The objects are streams of shared device batches; calls that arrive together run as one launch sequence (csrc/host/pool.cpp).
        if tracker in ("bytetrack", "sort", "ocsort") and not heavy:
            bt_sweep = {}
            try:
                Fb = min(F, zs + 30)
                for T in [t for t in (1, 16, 64, 256, 1024) if t <= S]:
                    dd = np.ascontiguousarray(host[:Fb, :T].transpose(1, 0, 2, 3))
                    L.pool_stats(reset=True)
                    fctx = L.Context(local)
                    fctx.lap_fast_stats(reset=True)
                    # 300 timed update() calls per object (the resident frames played back and forth), p50 / p99 of their latencies
                    res, _cs = L.bench_threads(tracker, dd, np.full((T, Fb), M, np.int32), warm=zs, device=local, frames=zs + 300)
                    ps_ = L.pool_stats()
                    fs_ = fctx.lap_fast_stats()  # a declined assignment goes to the exact kernel (2.3 ms at the north-star shape) and its whole round waits: that is the p99
                    declined_ = sum(v for k, v in fs_.items() if k.startswith("declined") or k in ("search_too_large", "certificate_arith", "too_many_tight", "not_unique"))
                    bt_sweep[f"T{T}"] = {"frames/s": res["frames_per_s"], "ms_per_update_p50": res["latency_ms_p50"], "ms_per_update_p99": res["latency_ms_p99"],
                                         "ms_per_update_mean": res["latency_ms_mean"], "ms_per_update_max": res["latency_ms_max"],
                                         "frames_timed": res["frames"], "launch_sequences": ps_["rounds"], "largest_round": ps_["max_round"],
                                         "assignments_fast_path": fs_["fast"], "assignments_declined_to_the_exact_kernel": declined_}
                bt_sweep["note"] = ("T objects of motcpp::trackers::" + {"bytetrack": "ByteTrack", "sort": "Sort", "ocsort": "OCSort"}[tracker] + " on T host threads, "
                                    "update(dets, img) with HOST detections (Eigen matrices; PCIe, the combiner's batching window and the copy of the result "
                                    f"table included), {zs} untimed frames first; launch_sequences = rounds the combiner ran for all frames of the leg; the p99 is decided by the rounds that "
                                    "wait for a declined assignment (DESIGN.md section 9)")
            except Exception as e:  # (diagnostic leg: never loses the line)
                bt_sweep["error"] = repr(e)
            sweep["basetracker_update"] = bt_sweep
        # the same call on ONE object whose lifecycle stays in a host stage machine (rounds 1-3; MOTCPP_LIFECYCLE=host), for comparison
        try:
            tk = L.Tracker(tracker, device=local)
            lat = []
            for f in range(zs + ks):
                d1 = host[f, 0]
                ta = time.perf_counter()
                tk.update(d1, embs[f, 0] if D else None)
                if f >= zs:
                    lat.append(time.perf_counter() - ta)
            lat = np.array(lat) * 1e3
            sweep["host_stage_machine_S1"] = {"frames/s": len(lat) / (lat.sum() * 1e-3), "ms_per_update_median": float(np.median(lat)), "ms_per_update_max": float(lat.max()),
                                              "steps": len(lat), "note": "one tracker object with the host stage machine of rounds 1-3 (C handle motcpp_tracker_create), "
                                              "update(dets) per frame, detections uploaded inside the call; the C++ classes use the pooled streams above"}
            tk.close()
        except Exception as e:  # (diagnostic leg: never loses the line)
            sweep["host_stage_machine_S1"] = {"error": repr(e)}
        sweep["note"] = ("synchronous steps of one batch of S streams (step = one frame of every stream; no sub-batches, one frame in flight), "
                         f"{zs} settling frames first; S = 1 is the latency of a single stream's frame on the device lifecycle")
    per_rank = None
    if world > 1:
        # every rank's own rate (frames of ITS streams / ITS wall time of the timed region), gathered over RCCL like everything else
        per_rank = gather_rank_rates(dist, torch, S * K / elapsed, S, f"cuda:{local}")  # backend "nccl" = RCCL
        te = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    frames = world * S * K
    value = frames / elapsed
    if long_run:
        long_run["vs_value"] = long_run["value"] * world / value
    # ---- roofline of the dominant kernel family (largest summed event time in the timed region) ----
    # ONE protocol: the dominant KERNEL — summed HIP-event time of its own launches on the launching stream inside the timed region
    # (composite families — the whole frame, the sparse + exact launch pairs — are listed under `kernels` but do not compete);
    # its average is what rocprofv3 --kernel-trace --stats of the same command shows for that kernel (profiles/*_kernel_stats_*.csv)
    # The dominant kernel of each workload is FIXED (chosen from the rocprofv3 kernel tables under profiles/, not from this run's event sums:
    # round 3 reported embed_kernel in one C3 run and the assignment composite in another): NS / C2 / C5 lap_sparse_kernel of the first
    # association, C3 embed_kernel<cosine> (bound: fp32 MFMA), C4 the exact lap_kernel of the first association, SORT its one assignment launch.
    composite = {"frame_all_kernels"} | ({"lap"} if "lap1_sparse" in stats else set())
    # (round 5: BoT-SORT's appearance term is evaluated for the pairs that pass the proximity test only — cosine_gated.hip — and the kernel table of
    # C3, profiles/r05f_kernel_stats_C3.csv, is led by feat_kernel, the appearance-feature maintenance: HBM-bound)
    fixed = {"NS": "lap1_sparse", "C2": "lap1_sparse", "C5": "lap1_sparse", "C3": "cosine" if DENSE_COS else "feat", "C4": "lap", "SORT": "lap"}.get(args.workload)
    fam = fixed if fixed in stats and stats[fixed]["launches"] else max((k for k in stats if k not in composite), key=lambda k: stats[k]["ms"])
    if not stats[fam]["launches"]:
        raise SystemExit(f"bench.py: the profiled kernel family {fam!r} reports 0 launches in the timed region (profiling events missing)")
    st = stats[fam]
    launches = max(st["launches"], 1)
    interval_ms = st["ms"] / launches  # HIP-event interval inside the timed region: with sub-batches on their own streams it starts when the
    # stream's previous kernel ends, i.e. it includes the wait for CUs the other streams hold — NOT the kernel's duration
    avg_ms = interval_ms
    duration_source = "HIP events around the kernel's launches inside the timed region (one sub-batch stream: nothing else shares the GPU)"
    if isolated and fam in isolated and PIPE > 1:
        # the kernel's own duration: the same launches (same trackers, the steps right behind the timed region) with the sub-batches stepped
        # one at a time, HIP events on the launching stream. This is the figure rocprofv3 --kernel-trace --stats of this same command
        # reports as the kernel's average (profiles/r04*_kernel_stats_*.csv: 299.9 us against 298.6 us here for lap_sparse_kernel at NS)
        avg_ms = isolated[fam]["avg_launch_ms"]
        duration_source = ("HIP events around the kernel's launches, sub-batches stepped one at a time right after the timed region (same trackers, full "
                           "sub-batches): the kernel's own begin-to-end time. rocprofv3's --stats average of this command mixes launch sizes (stream sweep) and "
                           "time-shared launches (three sub-batches); the like-for-like check is a one-sub-batch command whose last launches are the timed "
                           "ones (--streams 6144 --pipeline 1 --no-outputs-resident, no other legs): events 692 us against 686 us for the same 20 launches in "
                           "rocprofv3's trace (profiles/r06e_NS_one_subbatch_last20.txt, tools/last_launches_avg.py)")
    bytes_per_launch = st["bytes"] / launches
    if fam == "cosine" and st["flops"] > 0:
        achieved = st["flops"] / launches / (avg_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F32_PEAK_TFLOPS}
    else:
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    kernel_names = {"lap1_sparse": "lap_sparse_kernel (first association: pool of tracked + lost tracks x high-score detections)",
                    "lap": "lap_sparse_kernel + lap_kernel (assignment launches of a frame)", "cosine": "embed_kernel<cosine>" if DENSE_COS else "embed_gated_kernel",
                    "feat": "feat_kernel (normalise / set / blend the appearance features: three launches per frame)",
                    "kf_update": "kf_update_blocks_kernel (ByteTrack device lifecycle: block-form covariances, DESIGN.md section 2) / kf_update8_kernel (dense records)", "kf_predict_boxes": "kf_kernel (predict + boxes)", "kf_initiate": "kf_kernel (initiate)",
                    "ocsort_cost": "ocsort_kernel"}
    roof.update({"kernel": kernel_names.get(fam, fam), "family": fam, "avg_launch_ms": avg_ms, "launches": st["launches"],
                 "problems_per_launch": st["tasks"] / launches, "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": None,
                 "protocol": duration_source, "timed_region_event_interval_ms": interval_ms,
                 "note": "algorithmic bytes of an assignment = 24 B per row and column (boxes + score in, x/y out): the solver recomputes costs from "
                         "the boxes, no matrix exists; the kernel is latency/dependency-bound (augmenting-path search), its HBM fraction is "
                         "reported as measured, see DESIGN.md"})
    if fam == "feat":
        roof["note"] = ("algorithmic bytes of the appearance-feature maintenance = 4 D bytes per feature row read or written: every detection's row is normalised "
                        "(read + write), a new track's feature is set (read + write), a matched track's is blended with the detection's and renormalised (two reads "
                        "+ write); counted per launch on the device (mot_bot_profile_feat). avg_launch_ms averages the frame's three launches, as rocprofv3's "
                        "table does for feat_kernel")
    if fam == "cosine" and st["flops"] > 0:
        roof["note"] = ("flops of a cosine-distance launch = 2 n m D (the fp32 MFMA contraction; norms and the 1 - sim epilogue not counted) against the dense fp32 "
                        "MFMA peak; algorithmic_bytes_per_launch = 4 ((n + m) D + n m): feature rows in, distances out (the kernel is MFMA-bound, the bytes are "
                        "there for the traffic comparison)")
    if isolated and fam in isolated:  # the same family with the GPU to itself (see kernels_isolated)
        roof["isolated"] = isolated[fam]
    prof = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
    if os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            per_problem = (pj.get(fam) or pj.get("lap") or {}).get("hbm_bytes_per_problem")
            if per_problem is not None:  # PMC passes are separate runs (profiles/README.md); scaled to this run's problems per launch
                roof["traffic"] = per_problem * st["tasks"] / launches
                # counters are collected in their own runs: the file says which kernel sources it was collected for; a file from older sources
                # is still shown (the kernels' traffic rarely moves) but marked — round 5's line quoted round-4 counters without saying so
                roof["traffic_source"] = {"file": os.path.relpath(prof, ROOT), "tag": pj.get("tag") or None,
                                          "collected_for_kernel_sources": pj.get("kernel_sources_sha"), "this_run_kernel_sources": KSRC,
                                          "stale": pj.get("kernel_sources_sha") != KSRC}
        except Exception:
            pass
    # The same throughput priced with SURVEY §8(d)'s per-frame bytes — the formulation that materialises each stage's cost
    # matrix (written once, read once by the solver), nominal N = P tracks, K = M matches, first/biggest stage only, as in
    # its worked values (C2 0.49 MB, NS 4.9 MB). `achieved` above counts what the fused kernels need instead: the
    # assignment kernel recomputes costs from the boxes, so the 2·n·m·4 B term never exists in HBM.
    kf_d = 7 if tracker in ("sort", "ocsort") else 8
    survey_bytes = 4.0 * (2 * P * (kf_d + kf_d * kf_d) + 2 * M * (kf_d + kf_d * kf_d) + 4 * (P + M) + M + 2 * P * M + (P + M) * D + 8 * M)
    roof["survey_8d_equivalent"] = {"bytes_per_frame": survey_bytes, "GB/s_per_gpu": value / world * survey_bytes / 1e9,
                                    "frac": value / world * survey_bytes / 1e9 / HBM_PEAK_GBS,
                                    "note": "whole-job frames/s x SURVEY 8(d) bytes per frame (cost matrix materialised); not a kernel measurement"}
    # The fused pipeline's own byte model: what a frame HAS to move when no cost matrix is materialised —
    # detections in (24 B each), box-only prediction of the pool (52 B per track), one Kalman state read + written per match (round 6: 218 B — the
    # covariance in block form, 64 B, next to 32 B of mean; 604 B with 288-byte records), 24 B per row and column of every assignment (boxes + score in,
    # x / y out), 113 B per birth, 32 B per output row.
    if on_device and tracker == "bytetrack" and achieved_dims:
        fa, sa = achieved_dims["first_association"], achieved_dims["second_and_unconfirmed"]
        nfr = max(fa["problems"], 1)
        kfs = {k: stats[k]["tasks"] / nfr for k in ("kf_predict_boxes", "kf_initiate", "kf_update") if k in stats}
        lap_b = 24.0 * (fa["mean_tracks_N"] + fa["mean_dets_M"]) + 24.0 * (sa["mean_tracks_N"] + sa["mean_dets_M"]) * sa["problems"] / nfr
        rows_out = float(np.mean(cnt_all)) if len(cnt_all) else 0.0
        fb = 24.0 * M + 52.0 * kfs.get("kf_predict_boxes", 0.0) + KF_BT_UPDATE_B * kfs.get("kf_update", 0.0) + KF_BT_BIRTH_B * kfs.get("kf_initiate", 0.0) + lap_b + 32.0 * rows_out
        roof["fused_frame_model"] = {"bytes_per_frame": fb, "GB/s_per_gpu": value / world * fb / 1e9, "frac": value / world * fb / 1e9 / HBM_PEAK_GBS,
                                     "terms": {"detections": 24.0 * M, "predict_boxes": 52.0 * kfs.get("kf_predict_boxes", 0.0), "kalman_updates": KF_BT_UPDATE_B * kfs.get("kf_update", 0.0),
                                               "kalman_initiations": KF_BT_BIRTH_B * kfs.get("kf_initiate", 0.0), "assignments": lap_b, "output_rows": 32.0 * rows_out},
                                     "note": "whole-job frames/s x the bytes a frame of the fused pipeline must move (no N x M matrix exists); the path is "
                                             "latency / issue-bound, not bandwidth-bound, and this fraction says by how much"}
    import glob
    sq_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_sq_lap*.json")))  # newest round last (names carry the round)
    if fam in ("lap", "lap1_sparse") and sq_files:
        try:  # what actually bounds this kernel: instruction issue of the serial searches (SQ counters, separate PMC run)
            sq = json.load(open(sq_files[-1])).get(args.workload)
            if sq:
                roof["issue"] = {"stale": sq.get("kernel_sources_sha") != KSRC, "collected_for_kernel_sources": sq.get("kernel_sources_sha"),
                                 "wave_cycles_issuing_frac": sq["active_frac"], "wave_cycles_waiting_frac": sq["wait_any_frac"],
                                 "valu_insts_per_problem": sq["per_problem"]["SQ_INSTS_VALU"],
                                 "salu_insts_per_problem": sq["per_problem"]["SQ_INSTS_SALU"], "kernel": sq.get("kernel"),
                                 "source": os.path.relpath(sq_files[-1], ROOT)}
        except Exception:
            pass
    kernels = {k: {"ms_total": round(v["ms"], 3), "launches": v["launches"],
                   "GB/s": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 2) if v["ms"] > 0 else 0.0}
               for k, v in stats.items() if v["launches"]}
    for k, v in stats.items():  # the one contraction on the path: fp32 MFMA rate against the 157.3 TFLOP/s peak
        if v.get("flops", 0) > 0 and v["ms"] > 0:
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
            kernels[k].update({"TFLOP/s": round(tf, 2), "mfma_f32_frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)})
    gpu_busy_ms = stats["frame_all_kernels"]["ms"] if on_device else sum(v["ms"] for v in stats.values())

    # ---- CPU baseline: the oracle (CPU restatement of the reference path) on one core, stream 0 ----
    cpu = None
    parity = None
    if not args.no_cpu_baseline:  # (rank 0 only; the other ranks have returned)
        from tests import orclib
        orc = orclib.load()
        kind = {"sort": orclib.SORT, "bytetrack": orclib.BYTETRACK, "ocsort": orclib.OCSORT, "botsort": orclib.BOTSORT}[tracker]
        # parity check on frames the GPU path just processed: the first frames and the first timed ones of every sampled stream
        # (the oracle steps through every frame in between: its state has to be the tracker's); one oracle per stream, in parallel
        want = {f: kept[f] for f in range(n_kept_warm)}
        want.update({W + k: kept[n_kept_warm + k] for k in range(len(kept) - n_kept_warm)})
        last = max(want) if want else -1

        def check_stream(sid):
            to = orc.tracker(kind)
            bad = 0
            for fi in range(last + 1):
                oo = to.update(host[fi, sid], embs[fi, sid] if D else None)
                if fi in want and (oo.shape != want[fi][sid].shape or not np.array_equal(oo, want[fi][sid])):
                    bad += 1
            return bad
        with ThreadPoolExecutor(max(1, min(cpu_budget(), len(parity_ids)))) as ex:  # (the oracle call releases the GIL)
            per_stream = list(ex.map(check_stream, parity_ids))
        mism = int(sum(per_stream))
        # C3: the cosine distances are the one place where the reduction order is visible. Mode 0 above is the order the kernels compute (a
        # k-ordered fmaf chain = the fp32 MFMA); the reference's Eigen dot()/norm() in its -O2 / no -march build (CMakeLists.txt:231-236) is SSE2
        # lane sums without fused operations = the oracle's arith_mode 1 (oracle/orc_math.hpp::dot_chain). Same frames against THAT oracle: ids,
        # confidences, classes and detection indices identical, boxes within 1e-4 relative; and how far a cost has to move before any
        # assignment of the stream changes (tools/c3_margin_report.py's perturbation ladder on the problems of these very frames).
        ref_order = None
        if D:
            def check_stream_ref(sid):
                to = orc.tracker(kind)
                bad, worst = 0, 0.0
                for fi in range(last + 1):
                    oo = to.update(host[fi, sid], embs[fi, sid])
                    if fi in want:
                        g = want[fi][sid]
                        if oo.shape != g.shape or not np.array_equal(oo[:, 4:], g[:, 4:]):
                            bad += 1
                        elif oo.size:
                            rel = float(np.max(np.abs(oo[:, :4].astype(np.float64) - g[:, :4]) / np.maximum(np.abs(oo[:, :4]), 1.0)))
                            worst = max(worst, rel)
                            bad += int(rel > 1e-4)
                return bad, worst
            orc.set_arith_mode(1)
            try:
                with ThreadPoolExecutor(max(1, min(cpu_budget(), len(parity_ids)))) as ex:
                    per_ref = list(ex.map(check_stream_ref, parity_ids))
            finally:
                orc.set_arith_mode(0)
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:  # the problems of stream 0's first frames behind the settling, dumped by the oracle itself
                to = orc.tracker(kind)
                for fi in range(min(last + 1, 30)):
                    if fi == 20:
                        os.environ["ORC_LAP_DUMP"] = tmp
                    to.update(host[fi, parity_ids[0]], embs[fi, parity_ids[0]])
                os.environ.pop("ORC_LAP_DUMP", None)
                import glob as _glob
                import re as _re
                rng = np.random.default_rng(0)
                ladder = [1e-6, 1e-5, 1e-4, 1e-3, 1e-2]
                changed = {dlt: 0 for dlt in ladder}
                nprob = 0
                for fn in sorted(_glob.glob(os.path.join(tmp, "*.bin")))[:24]:
                    n_, m_ = map(int, _re.search(r"_(\d+)x(\d+)\.bin", fn).groups())
                    if n_ * m_ == 0:
                        continue
                    raw = np.fromfile(fn, np.float32)
                    th, cm = float(raw[0]), raw[1:].reshape(n_, m_)
                    x0, _y0 = orc.linear_assignment(cm, th)
                    nprob += 1
                    for dlt in ladder:
                        cp = (cm.astype(np.float64) + rng.uniform(-dlt, dlt, cm.shape)).astype(np.float32)
                        x1, _y1 = orc.linear_assignment(cp, th)
                        changed[dlt] += int(not np.array_equal(x0, x1))
            flipped = [dlt for dlt in ladder if changed[dlt]]
            ref_order = {"oracle_arith_mode": 1,
                         "what": "the same sampled stream-frames against the oracle with SSE-style four-lane mul/add dot products and the alternative Kalman "
                                 "factorisation orders (no fused operation in the cosine term: what matching.cpp:78-90 is under CMakeLists.txt:231-236)",
                         "stream_frames_checked": len(kept) * len(parity_ids),
                         "mismatching_stream_frames": int(sum(b for b, _w in per_ref)),
                         "criterion": "id, conf, cls, det_ind identical; boxes within 1e-4 relative (floor: one pixel)",
                         "max_rel_box_difference": max(w for _b, w in per_ref),
                         "assignment_margin": {"problems": nprob, "perturbation_ladder": ladder,
                                               "problems_whose_assignment_changed": {f"{dlt:g}": changed[dlt] for dlt in ladder},
                                               "smallest_perturbation_that_changed_an_assignment": (min(flipped) if flipped else None),
                                               "largest_perturbation_with_no_change": max([dlt for dlt in ladder if not changed[dlt] and (not flipped or dlt < min(flipped))], default=None),
                                               "note": "every cost of a problem moved by a uniform amount in [-delta, +delta], re-solved with the oracle's lapjv; a "
                                                       "reordered 256-term fp32 reduction moves a cosine distance by < 1e-6 (tests/test_gpu_reference_arith.py)"}}
        parity = {"streams_checked": len(parity_ids), "stream_ids": parity_ids, "sub_batches_covered": len({max(q for q in range(PIPE) if bounds[q] <= sid) for sid in parity_ids}),
                  "frames_checked_per_stream": len(kept), "stream_frames_checked": len(kept) * len(parity_ids), "mismatching_stream_frames": mism,
                  "mismatching_frames": mism, "streams_with_a_mismatch": int(sum(1 for b in per_stream if b)),
                  "criterion": "every output row bit for bit (array_equal)",
                  "oracle": "oracle/ in arith_mode 0 (the canonical summation orders, which the kernels reproduce bit for bit) - a CPU restatement of the "
                            "reference, pinned by the reference's own known answers only: N <= 3 assignments, IoU values, XYSR Kalman, SORT ids (DESIGN.md "
                            "section 5)" + ("; reference_order = the same frames against arith_mode 1 (SSE-style sums, no FMA)" if ref_order else "")}
        if ref_order:
            parity["reference_order"] = ref_order
        to2 = orc.tracker(kind)
        st0 = SynthStream(P, M, 1234, D)
        for _ in range(min(W, 40)):
            d, e = st0.next_frame()
            to2.update(d, e)
        n, tc = 0, 0.0
        while tc < args.cpu_seconds and n < 200000:
            d, e = st0.next_frame()
            ta = time.perf_counter()
            to2.update(d, e)
            tc += time.perf_counter() - ta
            n += 1
        cpu = {"value": n / tc, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{n} consecutive frames of stream 0 after {min(W, 40)} warm-up frames ({tc:.1f} s), oracle/ (scalar C++17, -O2)",
               "host_cores_available": os.cpu_count()}
        # the reference is single-threaded per tracker; "one tracker per core"  = independent oracle
        # processes, as many as this box's CPU budget, each on its own stream, all running at the same time; only the tracker calls are timed
        import subprocess
        nproc = cpu_budget()
        worker = os.path.join(ROOT, "tools", "cpu_baseline_worker.py")
        procs = [subprocess.Popen([sys.executable, worker, str(kind), str(P), str(M), str(D), str(5000 + i), str(W), str(args.cpu_seconds)],
                                  stdout=subprocess.PIPE, text=True) for i in range(nproc)]
        tot = 0.0
        for pr in procs:
            out = pr.communicate()[0].split()
            if pr.returncode == 0 and len(out) == 2:
                tot += int(out[0]) / float(out[1])
        cpu["all_cores"] = {"value": tot, "unit": "frames/s", "cores": nproc,
                            "sample": f"{nproc} oracle processes x {args.cpu_seconds:.0f} s, one stream each, concurrently; sum of per-process frames / time inside update()"}

    # C3: the one dense contraction of the path on the matrix cores. BoT-SORT's device lifecycle evaluates the appearance term for the pairs that pass
    # its proximity gate only (cosine_gated.hip: no MFMA launch in the frames timed above); the MFMA kernel is what DeepOC-SORT, StrongSORT and
    # utils::embedding_distance run — embed_kernel at this workload's shape (tracks x detections x D per camera, a sub-batch of cameras per launch),
    # timed here on device-resident features, next to the matrix-core busy counter of the same kernel from its own PMC run (profiles/).
    mfma = None
    if D and not args.no_cpu_baseline:
        try:
            import embed_microbench
            nt = max(1, min(512, S))
            r = embed_microbench.run(nt, P, M, D, reps=10)
            mfma = {"kernel": "embed_kernel<cosine> (mot_cosine_cost: fp32 MFMA v_mfma_f32_32x32x2_f32, 128 x 128 tiles)", "tasks_per_launch": nt,
                    "shape": [P, M, D], "ms_per_launch": r["ms_per_launch"], "TFLOP/s": r["TFLOP/s"], "peak_TFLOP/s": MFMA_F32_PEAK_TFLOPS,
                    "mfma_f32_frac": r["TFLOP/s"] / MFMA_F32_PEAK_TFLOPS, "flops": "2 n m D per task (norms and the 1 - sim epilogue not counted)",
                    "timing": "wall clock around 10 back-to-back launches between two stream synchronisations, after the timed region"}
            import glob as _g
            cf = sorted(_g.glob(os.path.join(ROOT, "profiles", "*pmc_mfma_embed*.json")))
            if cf:
                cj = json.load(open(cf[-1]))
                mfma["matrix_core_busy"] = {"SQ_VALU_MFMA_BUSY_CYCLES_over_GRBM_GUI_ACTIVE_x_SIMDs": cj.get("mfma_busy_frac"), "source": os.path.relpath(cf[-1], ROOT),
                                            "collected_for_kernel_sources": cj.get("kernel_sources_sha"), "stale": cj.get("kernel_sources_sha") != KSRC}
        except Exception as ex:  # noqa: BLE001
            mfma = {"error": repr(ex)}
    line = {
        "metric": "tracker.update() frames/sec at N_tracks x M_dets (assignment indices identical to the reference path)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W0,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc}", "tracker": tracker, "tracks": P, "dets_per_frame": M, "emb_dim": D,
                   "streams_per_gpu": S, "frames_per_step": world * S, "settle_frames": Z, "host_threads": ((1 if in_flight else PIPE) if on_device else threads), "sub_batches": PIPE, "frames_in_flight": 2 if in_flight else 1,
                   "sub_batch_stepping": ("free-running: every sub-batch's driver thread steps its own K frames, no barrier between sub-batches inside the timed region"
                                          if free_running else "lockstep"),
                   "lifecycle": "device (mot_bt_* / mot_sort_* / mot_bot_* / mot_oc_*: a fixed launch sequence per frame, no host decisions)" if on_device else "host stage machines",
                   "parallelism": f"{world} GPU(s) x {S} independent streams, lockstep stages",
                   "ranks": world, "rank_launcher": "torch.distributed.run (bench.py --gpus N starts it itself when no launcher set RANK)",
                   "table_gather": (args.gather + " (RCCL)") if world > 1 else args.gather,
                   "inputs": "detections (and embeddings) resident in HBM before the timed region; LAP arithmetic is f64/int32, Kalman/IoU f32"},
        "per_rank": per_rank,
        "roofline": roof, "cpu_baseline": cpu, "parity": parity, "mfma": mfma,
        "kernels": kernels,
        "lap_fast_path": fast_stats,
        "lap_behind_fast_path": diag_ctx.lap_behind_stats(),  # (whole run: the problems the exact kernel solved behind the sparse solver, cycle split of the slowest)
        "kernels_isolated": isolated,
        "long_run": long_run, "outputs_resident": outputs_resident, "stream_sweep": sweep,
        "achieved_problem_sizes": achieved_dims, "host_input": host_input,
        "gpu_busy_frac": gpu_busy_ms / (elapsed * 1e3) if elapsed > 0 else None,
        "flushes_per_step": (c1["flushes"] - c0["flushes"]) / K, "launches_per_step": (c1["launches"] - c0["launches"]) / K,
        "host_ms_per_step": {k: (c1[k] - c0[k]) / K for k in ("ms_begin", "ms_flush", "ms_advance", "ms_sync_wait")},
    }
    if args.trace_steps:
        marks = [t0] + step_marks[:K]
        line["step_ms"] = [round((marks[i + 1] - marks[i]) * 1e3, 3) for i in range(len(marks) - 1)]
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
