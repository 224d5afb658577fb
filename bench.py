#!/usr/bin/env python3
"""bench.py -- headline benchmark of the zzflate encoder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level L] [--mib M] [--gen text|random|log|mix]

A *step* is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
each rank compresses its own contiguous shard of packets (BASELINE.json configs[1]: 1 GiB of synthetic
"enwik-style" text per GPU, level 1, zlib container, 32 KiB packets) into a device buffer, then -- for
N > 1 -- the per-rank sizes/checksums are all-gathered and the compressed shards are gathered onto rank 0
with one grouped RCCL send/recv (the "single gather over xGMI" of the north star), straight into their final
offsets. That gather is left in flight while the next step is encoded into a second pair of buffers (--chunks 1,
default), or split into C pieces that travel while the next piece is encoded (--chunks C); the timed region
ends only when every gather has landed. `value` is whole-job input GB/s = bytes all ranks compressed /
max-over-ranks time. Scaling is weak (fixed work per GPU).

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  "roofline":     the encode kernel's algorithmic bytes (input read once + compressed bytes written once,
                  SURVEY.md 8d) over its mean launch duration measured with HIP events on the launch stream,
                  against the 8 TB/s HBM peak; "traffic" comes from profiles/traffic.json when present
  "cpu_baseline": the unmodified reference (oracle/_ref, kind "reference") or, if that build is absent, the
                  oracle restatement (kind "port"), timed on this box's host cores on a bounded sample.
  "check":        untimed validation of the last step: every packet inflated on the device and compared with its
                  input (zz_verify_last_device, all ranks), plus a 64 MiB prefix through zlib on rank 0.
The oracle is only the checker / baseline here, never the thing measured.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GEN = {"text": 0, "random": 1, "log": 2, "mix": 3}
SEEDS = {"text": 0x5EED0002, "random": 0x5EED0003, "mix": 0x5EED0004, "log": 0x5EED0005}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def effective_cores():
    """Cores this process may really use: the scheduler affinity mask, cut down by the cgroup CPU quota when there is one
    (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us). Returns (cores, affinity, quota or None)."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    cores = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return cores, affinity, quota


def cpu_baseline(sample: bytes, level: int, fmt: int, P: int = 32768):
    """Time the reference's CPU encoder on this box's host cores, in two forms:
      * whole-slice: one ZzFlateEncode(threaded=false) call per 16 MiB slice -- the reference as its own callers run it;
      * packet mode: the packet recipe (zzflate.cpp:101-125) over the same P-byte ranges the GPU uses -- the same work
        as the GPU's, and the same bytes: its ratio must equal the GPU line's.
    Each form is timed on ONE thread first and then on every core the process may use (affinity cut down by the cgroup
    quota), by a native driver (oracle/ref_harness.cpp zzref_bench: native threads, static partition, per-thread output
    buffers; no Python in the loop), so that `value / cores` can be held against the one-thread rate.
    Returns the cpu_baseline JSON object (`value` = whole-slice GB/s on all cores, `packet_mode` = the second form)."""
    cores, affinity, quota = effective_cores()
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libzzref.so")
    slice_bytes = 16 << 20
    nslices = max(1, len(sample) // slice_bytes)
    u64, u32, ci, vp, dbl = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_double
    if os.path.exists(ref_path):
        fn = ctypes.CDLL(ref_path).zzref_bench
        kind = "reference"
    else:
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        fn = ctypes.CDLL(os.path.join(ROOT, "oracle", "libzzoracle.so")).zzo_bench
        kind = "port"
    fn.restype = dbl
    fn.argtypes = [ci, vp, u64, u64, u64, ci, ci, ci, u32, ctypes.POINTER(u64), ctypes.POINTER(dbl)]
    buf = ctypes.create_string_buffer(sample, len(sample) + 64)   # >= 8 readable bytes after the data
    base = ctypes.addressof(buf)
    total_in = nslices * slice_bytes
    hl, tl = {0: (2, 4), 1: (10, 8), 2: (0, 0)}[fmt]

    def run(mode, threads, nitems):
        produced = (u64 * nslices)()
        secs = (dbl * threads)()
        wall = fn(mode, base, nslices, slice_bytes, nitems, threads, fmt, level, P, produced, secs)
        return nitems * slice_bytes / wall / 1e9, wall, list(produced), list(secs)

    out = {}
    for mode, name in ((0, "whole"), (1, "packets")):
        # one thread: ~4 slices (0.2-0.5 s at level 1); all cores: every slice at least once, every thread 24 slices
        # (1-3 s of work each at level 1, more at level 2): ~10-30 s of CPU work per form
        one, one_wall, _, _ = run(mode, 1, min(4, nslices))
        per_thread = 24
        nitems = max(nslices, per_thread * cores)
        allv, wall, produced, secs = run(mode, cores, nitems)
        eff = (allv / cores) / one if one > 0 else None
        out[name] = {
            "value": round(allv, 4), "unit": "GB/s", "cores": cores, "kind": kind,
            "one_thread": round(one, 4), "per_core": round(allv / cores, 4),
            "scaling_efficiency": round(eff, 3) if eff is not None else None,
            "scaling_flag": ("per-core rate is less than half the one-thread rate: memory bandwidth, SMT siblings or a CPU "
                             "quota the affinity mask does not show; effective cores = value / one_thread = "
                             f"{allv / one:.1f}") if (eff is not None and eff < 0.5) else None,
            "thread_seconds_min_max": [round(min(secs), 2), round(max(secs), 2)], "wall_seconds": round(wall, 2),
            "ratio": round((sum(produced) + (hl + tl if mode else 0)) / total_in, 4),
            "items": nitems,
        }
    res = dict(out["whole"])
    res["affinity_cores"] = affinity
    res["cgroup_cpu_quota"] = quota
    res["sample"] = (f"{out['whole']['items']} x {slice_bytes >> 20} MiB slices ({nslices} distinct) of the same input, level {level}, "
                     f"one ZzFlateEncode(threaded=false) call per slice, native threads with a static partition "
                     f"(oracle/ref_harness.cpp zzref_bench), {cores} threads; one thread alone on {min(4, nslices)} slices first")
    pk = out["packets"]
    pk["sample"] = (f"the same slices as {P}-byte packets (the recipe of zzflate.cpp:101-125, one Encoder per packet): the GPU's "
                    f"work and the GPU's bytes")
    res["packet_mode"] = pk
    return res


def summarize_regions(times, steps):
    """The contract's timed region -- exactly `steps` steps between two barriers -- measured `len(times)` times: the line's
    ms_per_step (and `value`) come from the MEDIAN region, so that a short --steps (the driver passes 20: 0.15 s) does not hang
    the headline on one sample of the clock; min and max say how far the regions spread. Pure arithmetic (tests/test_bench_host.py)."""
    ts = sorted(times)
    k = len(ts)
    med = ts[k // 2] if k % 2 else 0.5 * (ts[k // 2 - 1] + ts[k // 2])
    return {"repeats": k, "region_seconds_median": med, "ms_per_step": med / steps * 1e3,
            "ms_per_step_min": ts[0] / steps * 1e3, "ms_per_step_max": ts[-1] / steps * 1e3, "timed_seconds_total": sum(ts)}


def want_another_region(times, min_total=1.0, max_repeats=64):
    """keep timing regions until they add up to a second (at least one, at most 64)"""
    return not times or (len(times) < max_repeats and sum(times) < min_total)


def scalar_bound(scalar_insts, kernel_ms, cus=256, clock_ghz=2.4):
    """roofline.scalar: a CU has ONE scalar unit, one instruction per cycle, for all its wavefronts (tools/ubench_scalar.hip:
    0.92 per cycle measured with 32 waves on it). (SALU + branch instructions of a launch) / (CUs x clock x kernel time) = the
    unit's mean issue rate per CU per cycle; `frac` is that against the measured ceiling."""
    if not scalar_insts or not kernel_ms or kernel_ms <= 0:
        return None
    per_cycle = scalar_insts / (cus * clock_ghz * 1e9 * kernel_ms * 1e-3)
    ceiling = 0.92
    return {"bound": "scalar unit", "scalar_instructions": scalar_insts, "per_cu_per_cycle": round(per_cycle, 4), "ceiling_per_cu_per_cycle": ceiling,
            "frac": round(per_cycle / ceiling, 4), "bound_ms": round(scalar_insts / (cus * clock_ghz * 1e9 * ceiling) * 1e3, 3),
            "clock_ghz": clock_ghz, "cus": cus}


def launch_ranks(n):
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment,
    as torch.distributed.run would set them) and wait for them. Children are separate processes started from a process
    that never initialised the GPU; nothing is re-executed in place."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC): documented by the build/run environment of this project as required for
        # RCCL and device-memory sharing across processes on this pool's host driver, and exported there already; only set
        # when the caller's environment does not say otherwise (never overridden). DESIGN.md 6.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # first contact with a multi-GPU node: whatever RCCL has to say about a failing rank goes to that rank's stderr, which is
        # this process's (inherited)
        env.setdefault("NCCL_DEBUG", "WARN")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    try:
        while any(pr.poll() is None for pr in procs):
            for pr in procs:
                rc = pr.poll()
                if rc not in (None, 0) and not worst:
                    worst = rc
                    print(f"bench.py: rank {procs.index(pr)} exited with code {rc}; its stderr (NCCL_DEBUG={os.environ.get('NCCL_DEBUG', 'WARN')}) is above; "
                          "stopping the other ranks", file=sys.stderr, flush=True)
                    for other in procs:                # a rank died: the others would wait in a collective forever
                        if other.poll() is None:
                            other.terminate()
            time.sleep(0.2)
        for pr in procs:
            worst = worst or pr.returncode
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)      # (a timed region; regions are repeated until they add up to a second)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--mib", type=int, default=1024, help="input MiB per GPU")
    ap.add_argument("--gen", default="text", choices=list(GEN))
    ap.add_argument("--format", default="zlib", choices=["zlib", "gzip", "deflate"])
    ap.add_argument("--packet", type=int, default=32768)
    ap.add_argument("--chunks", type=int, default=1, help="N > 1: 1 = one launch and one gather per step, the gather left in "
                    "flight under the next step's encoding (double-buffered); C > 1 = C pieces per shard, piece c "
                    "travels to rank 0 while piece c+1 is encoded")
    ap.add_argument("--warm", type=int, default=0, help="level 1: warm window in bytes (0 = cold packets = the reference's threaded "
                    "stream, the headline; > 0 is the beyond-reference mode of SURVEY.md 8f.3)")
    ap.add_argument("--inflight", type=int, default=1, choices=[1, 2], help="one GPU: calls in flight. 1 (default): every step is one "
                    "synchronous zz_encode_device call, and the HIP-event kernel time is the kernel's own. 2: two contexts on two "
                    "streams (zz_encode_device_async / zz_encode_finish), the next step's encode kernel fills the CUs the previous "
                    "one's last packets leave idle; per-kernel event times then include that sharing, so roofline is not reported")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "none"], help="N > 1: 'rccl' (default) = the north star's path, every "
                    "step's compressed shards are gathered onto rank 0; 'none' = encode only, nothing is exchanged (how the "
                    "encoders alone scale). The default line carries the encode-only rate too (`exchange.encode_only`)")
    ap.add_argument("--gather-root", default="0", choices=["0", "rotate"], help="N > 1: '0' (default) = every step's stream is gathered "
                    "onto rank 0; 'rotate' = step i's onto rank i mod N (still ONE grouped send/recv per stream), so that a run of "
                    "streams is not capped by ONE GPU's seven inbound xGMI links")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="the timed region (exactly --steps steps) is repeated until the regions "
                    "add up to this many seconds; 0 = one region (profiler passes)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the level-2 side measurement")
    ap.add_argument("--no-sequential", action="store_true", help="skip the threaded=false side measurement (64 MiB on one wavefront: seconds)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: become one. This process has not touched the GPU (torch is not even imported yet); it
        # starts one fresh child per rank -- the analogue of the reference's in-process fan-out, zzflate.cpp:127-132 --
        # and exits with the worst of their exit codes. Rank 0's JSON line goes to this process's stdout.
        sys.exit(launch_ranks(args.gpus))

    import torch
    import zzflate_amd as zz

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does "
              f"that itself; python -m torch.distributed.run --nproc-per-node N bench.py --gpus N works too)", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # ZZ_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: every rank uses GPU
    # (LOCAL_RANK mod device_count) and the exchange runs over gloo on host copies of the shards
    backend = os.environ.get("ZZ_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    # ZZ_BENCH_FORCE_DIST=1: take the distributed path even with one rank (RCCL init, all-gather, all-reduce, barrier
    # and the assembly on rank 0 run on a one-GPU box; the send/recv of the gather needs a second GPU)
    multi = world > 1 or os.environ.get("ZZ_BENCH_FORCE_DIST") == "1"
    dev = local % ndev if world > 1 else 0
    if multi:
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev)
    fmt = {"zlib": 0, "gzip": 1, "deflate": 2}[args.format]
    ctx = zz.Context(dev)
    ctx.enable_timing(True)
    if args.warm:
        ctx.set_warm_window(args.warm)
    if args.level > 3:
        ctx.set_extended_levels(True)      # levels 4..6: beyond the reference (chains, lazy matching, package-merge), SURVEY.md 8f.2

    n = args.mib << 20                 # bytes per rank (weak scaling)
    total_n = n * world
    P = args.packet
    # rank r holds bytes [r*n, (r+1)*n) of the job's input, plus (for r > 0) the 64 KiB in front of them:
    # level >= 2 backward match extension may read up to 258 bytes before the shard (zz_encode_shard_device halo)
    halo = 65536 if rank > 0 else 0
    buf = torch.empty(halo + n + 64, dtype=torch.uint8, device="cuda")
    ctx.generate(GEN[args.gen], SEEDS[args.gen], rank * n - halo, buf, halo + n)
    src = buf[halo:]
    cap = zz.bound(n, 2, min(args.level, 3), P)
    shard = torch.empty(cap, dtype=torch.uint8, device="cuda")
    xdev = "cuda" if backend == "nccl" else "cpu"     # where the exchange buffers live
    rotate = multi and args.gather_root == "rotate"
    holds = multi and (rank == 0 or rotate)             # this rank is the gather target of some steps
    gathered = torch.empty(cap * world + 32, dtype=torch.uint8, device=xdev) if holds else None
    # step-overlapped mode: a second pair of buffers, so that step i's gather can still be in flight during step i+1
    shard_b = [shard, torch.empty(cap, dtype=torch.uint8, device="cuda") if multi else None]
    gathered_b = [gathered, torch.empty(cap * world + 32, dtype=torch.uint8, device=xdev) if holds else None]
    root_b = [0, 0]                                     # the gather target of the stream in each pair of buffers
    pending = [None, None]
    # N > 1, --chunks 1: two contexts on two streams, alternating, so that step i's shard can be enqueued BEFORE step i-1's
    # size and checksum are exchanged (zz_encode_shard_device_async / zz_encode_shard_finish): the host-side part of the
    # exchange (all-gather of the triples, reading them, posting the send/recv) then runs under step i's kernels
    mctx = [ctx, None]
    mstream = [None, None]
    if multi:
        mctx[1] = zz.Context(dev)
        mctx[1].enable_timing(True)
        if args.warm:
            mctx[1].set_warm_window(args.warm)
        if args.level > 3:
            mctx[1].set_extended_levels(True)
        mstream = [torch.cuda.Stream(), torch.cuda.Stream()]
    enq = [False, False]
    C = max(1, args.chunks) if multi else 1
    assert n % (C * P) == 0, "--mib must split into --chunks packet-aligned pieces"
    pipe = None
    if multi and C > 1:
        from zzflate_amd import sharded
        pipe = sharded.PipelinedGather(dist, fmt, cap, torch.device(xdev))
    torch.cuda.synchronize()

    kernel_ms = []
    state = {}
    # One GPU, --inflight 2: two calls in flight on two streams (zz_encode_device_async / zz_encode_finish, two contexts,
    # two output buffers): step i+1's encode kernel fills the CUs that step i's last packets leave idle and runs under
    # step i's compaction, checksum fold and result copy. Every step is a complete call; the timed region ends when all
    # are done. --inflight 1 (default): the same code with one lane, i.e. one synchronous call per step.
    lanes = []
    if (not multi):
        for b in range(args.inflight):
            c2 = ctx if b == 0 else zz.Context(dev)
            if b:
                c2.enable_timing(True)
                if args.warm:
                    c2.set_warm_window(args.warm)
                if args.level > 3:
                    c2.set_extended_levels(True)
            lanes.append({"ctx": c2, "stream": torch.cuda.Stream(), "dst": shard if b == 0 else torch.empty(cap, dtype=torch.uint8, device="cuda"),
                          "busy": False})

    def collect(b):
        ln = lanes[b]
        if ln["busy"]:
            w = ln["ctx"].finish()
            ln["busy"] = False
            kernel_ms.append(ln["ctx"].last_kernel_ms())
            state["out_bytes"] = w
            state["comp_bytes"] = w
            state["last_lane"] = b

    def post_gather(b):
        """finish the shard enqueued on pair b and start its exchange: sizes/checksums all-gather + ONE grouped send/recv of
        the compressed shards to rank 0, left in flight"""
        if not enq[b]:
            return
        from zzflate_amd import sharded
        enq[b] = False
        w, cks = mctx[b].finish_shard(fmt)
        kernel_ms.append(mctx[b].last_kernel_ms())
        state["last_buf"] = b
        state["comp_bytes"] = w
        state["last_cks"] = cks
        if args.gather == "none":
            state["out_bytes"] = w * world                  # (nothing assembled: the ratio is this rank's)
            return
        xshard = shard_b[b] if backend == "nccl" else shard_b[b][:w].cpu()
        pending[b] = sharded.gather_stream(dist, fmt, xshard, w, cks, n, gathered_b[b] if rank == root_b[b] else None, wait=False, root=root_b[b])

    def step():
        if (not multi):
            b = state.get("step", 0) % len(lanes)
            state["step"] = state.get("step", 0) + 1
            collect(b)                                      # the call that used this context before
            ln = lanes[b]
            ln["ctx"].encode_async(src, n, ln["dst"], cap, fmt, args.level, P, stream=ln["stream"].cuda_stream)
            ln["busy"] = True
            if len(lanes) == 1:
                collect(b)                                  # one call at a time: wait for it, as zz_encode_device does
            return
        elif pipe is None:
            b = state.get("step", 0) & 1
            state["step"] = state.get("step", 0) + 1
            if pending[b] is not None:                      # the gather that used this pair of buffers two steps ago
                tot = pending[b].wait()
                pending[b] = None
                if rank == 0:
                    state["out_bytes"] = tot
            root_b[b] = ((state["step"] - 1) % world) if rotate else 0
            # (a request's wait() orders the CURRENT stream behind the transfer; the encode runs on a stream of its own and
            # must not overwrite the shard buffer while that send may still be reading it)
            mstream[b].wait_stream(torch.cuda.current_stream())
            mctx[b].encode_shard_async(src, n, shard_b[b], cap, halo=halo, is_last=(rank == world - 1), checksum=fmt,
                                       level=args.level, packet_size=P, stream=mstream[b].cuda_stream)
            enq[b] = True
            post_gather(b ^ 1)                              # the previous step's exchange, under this step's kernels
            return
        else:
            # the shard in C pieces: piece c is on its way to rank 0 (async grouped send/recv) while c+1 is encoded
            pipe.begin()
            pn, wsum, kms, off = n // C, 0, 0.0, 0
            for c in range(C):
                piece = shard[off:]
                w, cks = ctx.encode_shard(src[c * pn:], pn, piece, cap - off, halo=halo + c * pn,
                                          is_last=(rank == world - 1 and c == C - 1), checksum=fmt,
                                          level=args.level, packet_size=P)
                kms += ctx.last_kernel_ms()
                pipe.push(piece if backend == "nccl" else piece[:w].cpu(), w, cks, pn)
                off += (w + 255) & ~255
                wsum += w
            tot = pipe.finish(gathered)
            if rank == 0:
                state["out_bytes"] = tot
            state["comp_bytes"] = wsum
            kernel_ms.append(kms)
            return

    def barrier():
        for b in range(len(lanes)):                         # every call in flight is finished before the clock is read
            collect(b)
        for b in (0, 1):                                    # the last shard enqueued: finish it, exchange it
            post_gather(b)
        for b in (0, 1):                                    # every gather has landed before the clock is read
            if pending[b] is not None:
                tot = pending[b].wait()
                pending[b] = None
                if rank == 0:
                    state["out_bytes"] = tot
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms.clear()
    # The timed region: EXACTLY --steps steps between two barriers (+ synchronize), MAX over ranks -- and that region again until
    # the regions add up to a second (every rank sees the same reduced times, so every rank stops at the same repeat).
    region_times = []
    while want_another_region(region_times, args.min_seconds):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if multi:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        region_times.append(dt)
    regions = summarize_regions(region_times, args.steps)
    dt = regions["region_seconds_median"]

    # ---- validation of what was just measured (untimed) ----------------------------------------------------------
    # every packet this rank produced in the last step is inflated on the device and compared with its input
    # (zz_verify_last_device: a plain RFC 1951 decoder, one lane per packet); rank 0 also inflates a prefix with zlib
    check = {}
    torch.cuda.synchronize()
    tv = time.perf_counter()
    # ---- N > 1: what the exchange costs, measured apart from the overlapped run above (untimed for `value`) ------------
    # encode_only: the same steps with nothing exchanged; gather_ms_serial: the exchange of the last step's shards on its own,
    # start to landed, nothing else running; bytes_into_rank0: what rank 0's inbound links carry per step. With these a
    # scaling run can tell "the encoders do not scale" from "rank 0's links are full" (DESIGN.md 6).
    exchange = None
    if multi and pipe is None and args.gather == "rccl":
        from zzflate_amd import sharded
        k2 = max(2, min(args.steps, 5))
        args.gather = "none"
        step(); barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        barrier()
        d1 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(d1, op=dist.ReduceOp.MAX)
        args.gather = "rccl"
        b = state["last_buf"]
        w, cks = state["comp_bytes"], state["last_cks"]
        xshard = shard_b[b] if backend == "nccl" else shard_b[b][:w].cpu()
        gms, tot = [], None
        for _ in range(k2 + 1):
            dist.barrier(); torch.cuda.synchronize()
            t2 = time.perf_counter()
            tot = sharded.gather_stream(dist, fmt, xshard, w, cks, n, gathered_b[b], wait=True)
            torch.cuda.synchronize()
            gms.append((time.perf_counter() - t2) * 1e3)
        g1 = torch.tensor([sum(gms[1:]) / k2], dtype=torch.float64, device="cuda")      # (the first one warms the connections up)
        dist.all_reduce(g1, op=dist.ReduceOp.MAX)
        # the same exchange onto the LAST rank: with --gather-root rotate every rank takes its turn as the target
        g2 = None
        if rotate:
            gms2 = []
            for _ in range(k2 + 1):
                dist.barrier(); torch.cuda.synchronize()
                t2 = time.perf_counter()
                sharded.gather_stream(dist, fmt, xshard, w, cks, n, gathered_b[b] if rank == world - 1 else None, wait=True, root=world - 1)
                torch.cuda.synchronize()
                gms2.append((time.perf_counter() - t2) * 1e3)
            g2 = torch.tensor([sum(gms2[1:]) / k2], dtype=torch.float64, device="cuda")
            dist.all_reduce(g2, op=dist.ReduceOp.MAX)
            tot = sharded.gather_stream(dist, fmt, xshard, w, cks, n, gathered_b[b], wait=True)      # rank 0 holds the last stream again (validation)
        if rank == 0:
            state["out_bytes"] = tot
            hl_, tl_ = {0: (2, 4), 1: (10, 8), 2: (0, 0)}[fmt]
            inbound = tot - hl_ - tl_ - w
            e_ms = float(d1.item()) / k2 * 1e3
            exchange = {"gather": "rccl" if backend == "nccl" else backend,
                        "encode_only": {"value": round(total_n / (e_ms * 1e-3) / 1e9, 3), "unit": "GB/s", "ms_per_step": round(e_ms, 3), "steps": k2},
                        "gather_root": args.gather_root,
                        "gather_ms_serial": round(float(g1.item()), 3),
                        "gather_ms_serial_onto_last_rank": round(float(g2.item()), 3) if g2 is not None else None,
                        "bytes_into_rank0_per_step": inbound,
                        "inbound_GBps_serial": round(inbound / (float(g1.item()) * 1e-3) / 1e9, 2) if inbound else 0.0}
    vctx = lanes[state.get("last_lane", 0)]["ctx"] if (not multi) else (mctx[state.get("last_buf", 0)] if pipe is None else ctx)   # the context of the last step
    bad, first_bad = vctx.verify_last()
    tv = time.perf_counter() - tv
    if multi:
        tb = torch.tensor([bad], dtype=torch.int64, device="cuda")
        dist.all_reduce(tb)
        bad = int(tb.item())
    check["device_inflate"] = {"packets": ((n + P - 1) // P) * world, "bad": bad, "seconds_rank0": round(tv, 3)}
    if rank == 0:
        import zlib
        if multi and args.gather == "none":                 # nothing was assembled: rank 0's own shard, raw DEFLATE
            out_t = shard_b[state.get("last_buf", 0)]
            k = min(state["comp_bytes"], 96 << 20)
            wb = -15
        else:
            out_t = lanes[state.get("last_lane", 0)]["dst"] if (not multi) else gathered_b[state.get("last_buf", 0)]
            k = min(state["out_bytes"], 96 << 20)
            wb = {0: 15, 1: 31, 2: -15}[fmt]
        head = out_t[:k].cpu().numpy().tobytes()
        o = zlib.decompressobj(wb)
        try:
            dec = o.decompress(head, 64 << 20)
        except zlib.error:            # an invalid stream is reported in the line (inflate_prefix_ok false), not as a crash
            dec = b""
        ref = src[:len(dec)].cpu().numpy().tobytes() if len(dec) <= n else None
        check["inflate_prefix_ok"] = bool(ref is not None and dec == ref)
        check["inflate_prefix_bytes"] = len(dec)

    extra = {}
    if rank == 0 and (not multi) and not args.no_extra and args.level == 1:
        # the config's "dynamic Huffman" wording means reference level 2: side measurement
        try:
            cap2 = zz.bound(n, fmt, 2, P)
            dst2 = torch.empty(cap2, dtype=torch.uint8, device="cuda")
            w2 = ctx.encode(src, n, dst2, cap2, fmt, 2, P)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                w2 = ctx.encode(src, n, dst2, cap2, fmt, 2, P)
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t2) / reps
            extra["level2"] = {"value": round(n / d2 / 1e9, 3), "unit": "GB/s", "ratio": round(w2 / n, 4),
                               "kernel_ms": round(ctx.last_kernel_ms(), 3)}
            del dst2
        except Exception as e:   # level 2 not available yet
            extra["level2"] = {"error": str(e)}
        # BASELINE configs[3] words its quality level as "level 6 (longer hash chains)": the extended level 6 (beyond the
        # reference: chains of depth 8, lazy matching, package-merge) on the same input, as a side measurement
        try:
            ctx6 = zz.Context(dev)
            ctx6.set_extended_levels(True)
            ctx6.enable_timing(True)
            cap6 = zz.bound(n, fmt, 2, P)
            dst6 = torch.empty(cap6, dtype=torch.uint8, device="cuda")
            w6 = ctx6.encode(src, n, dst6, cap6, fmt, 6, P)
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            for _ in range(2):
                w6 = ctx6.encode(src, n, dst6, cap6, fmt, 6, P)
            torch.cuda.synchronize()
            d6 = (time.perf_counter() - t6) / 2
            bad6, _ = ctx6.verify_last()
            extra["level6"] = {"value": round(n / d6 / 1e9, 3), "unit": "GB/s", "ratio": round(w6 / n, 4),
                               "kernel_ms": round(ctx6.last_kernel_ms(), 3), "device_inflate_bad": bad6,
                               "note": "extended level, not a reference stream: bit-exact with the oracle's definition (tests), inflated on the device here"}
            del dst6, ctx6
        except Exception as e:
            extra["level6"] = {"error": str(e)}

        # The reference's threaded=false form (zzflate.cpp:84-95: ONE Encoder over the whole input, what all of its own tests
        # call) is one dependency chain per call, so it runs on one wavefront: a compatibility mode. Its rate, stated: one call
        # of 64 MiB at levels 1 and 2, and 64 concurrent callers of 1 MiB each (64 contexts, 64 host threads) to show that
        # independent callers scale over the CUs.
        try:
            if args.no_sequential:
                raise RuntimeError("skipped (--no-sequential)")
            import threading
            seq = {}
            nseq = min(n, 64 << 20)
            capq = max(zz.bound(nseq, fmt, 2, P), 2 * nseq)
            dstq = torch.empty(capq, dtype=torch.uint8, device="cuda")
            for lvl in (1, 2):
                ctx.encode_stream(src, 1 << 20, dstq, capq, fmt, lvl)
                torch.cuda.synchronize(); tq = time.perf_counter()
                wq = ctx.encode_stream(src, nseq, dstq, capq, fmt, lvl)
                torch.cuda.synchronize(); dq = time.perf_counter() - tq
                seq[f"level{lvl}_one_call"] = {"value": round(nseq / dq / 1e9, 4), "unit": "GB/s", "bytes": nseq, "ratio": round(wq / nseq, 4)}
            K, piece = 64, 1 << 20
            ctxs = [zz.Context(dev) for _ in range(K)]
            outs = [torch.empty(2 * piece + 4096, dtype=torch.uint8, device="cuda") for _ in range(K)]
            strs = [torch.cuda.Stream(device=dev) for _ in range(K)]
            def one(i, lvl):
                ctxs[i].encode_stream(src[i * piece:(i + 1) * piece], piece, outs[i], 2 * piece + 4096, fmt, lvl, stream=strs[i].cuda_stream)
            for lvl in (1, 2):
                for rep in range(2):             # (the first round allocates the contexts' workspaces)
                    torch.cuda.synchronize(); tq = time.perf_counter()
                    th = [threading.Thread(target=one, args=(i, lvl)) for i in range(K)]
                    for t_ in th: t_.start()
                    for t_ in th: t_.join()
                    torch.cuda.synchronize(); dq = time.perf_counter() - tq
                seq[f"level{lvl}_64_callers_of_1MiB"] = {"value": round(K * piece / dq / 1e9, 4), "unit": "GB/s"}
            seq["note"] = "threaded=false (the reference's single Encoder): one wavefront per call, a compatibility mode; packet mode (threaded=true) is `value`"
            extra["sequential"] = seq
            del ctxs, outs, dstq
        except Exception as e:
            extra["sequential"] = {"error": str(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:        # (the CPU leg runs at N = 1 only: the other ranks would wait for it)
        sample_bytes = min(n, 1 << 30)
        # the reference has no level above 3 (zzflate.cpp:201,230): its level 3 is the baseline of the extended levels
        cpu = cpu_baseline(src[:sample_bytes].cpu().numpy().tobytes(), min(args.level, 3), fmt, P)

    if rank == 0:
        ms = dt / args.steps * 1e3
        kms = sum(kernel_ms) / max(1, len(kernel_ms))
        comp = state["comp_bytes"]
        algo_bytes = n + comp                      # per launch on this rank: input once + compressed once
        achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else None
        if (not multi) and len(lanes) > 1:
            achieved = None                        # event times of overlapping kernels include the sharing of the CUs
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/collect_profiles.sh; counters
        # cannot be read inside this run). profiles/traffic.json records which sources it was measured on: it is only
        # reported while those are the sources this run was built from.
        traffic, traffic_src, insts, tj_scalar = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                import hashlib
                tj = json.load(open(tpath))
                hsh = hashlib.sha256()
                cdir = os.path.join(ROOT, "zzflate_amd", "csrc")
                for f in sorted(os.listdir(cdir)):
                    hsh.update(f.encode()); hsh.update(open(os.path.join(cdir, f), "rb").read())
                if tj.get("source_sha256") == hsh.hexdigest():
                    traffic = tj.get(f"level{args.level}_{args.gen}_{args.mib}MiB")
                    insts = tj.get(f"instructions_level{args.level}_{args.gen}_{args.mib}MiB")
                    tj_scalar = tj.get(f"scalar_instructions_level{args.level}_{args.gen}_{args.mib}MiB")
                    traffic_src = f"profiles/traffic.json @ {tj.get('git_sha')}"
                    # the counters were collected on the default (cold, two-parser) kernels: another kernel's line does not borrow them
                    other = bool(args.warm) or os.environ.get("ZZFLATE_L1_KERNEL") == "classic" or os.environ.get("ZZFLATE_L2_KERNEL") == "classic" \
                        or (args.level == 1 and hasattr(zz.lib, "zz_debug_l1_kernel") and zz.lib.zz_debug_l1_kernel(ctx._h) != 2)
                    if other and traffic is not None:
                        traffic, insts, tj_scalar = None, None, None
                        traffic_src += " was measured on the default kernel of this level, not on the one this run launched: not reported"
                else:
                    traffic_src = f"profiles/traffic.json @ {tj.get('git_sha')} is stale (kernel sources changed since): not reported"
            except Exception:
                traffic = None
        # The bound that applies to levels >= 1 is not HBM but instruction issue (SURVEY.md hard part 5): a
        # wavefront issues at most one instruction per ~5 cycles (tools/ubench_valu.hip), and the LDS hash table admits 9
        # workgroups of 2 wavefronts per CU. issue.bound_ms = the kernel's wave-level instructions (SQ_INSTS_*, from the same
        # profile pass as `traffic`) x 5 cycles / (resident wavefronts x clock): what the kernel would take if every resident
        # wavefront issued flat out; frac = bound_ms / kernel_ms.
        issue = None
        # which level-1 kernel ran: asked of the library (ZZFLATE_L1_KERNEL=classic and a negative LDS-order verdict both give k_encode_l1)
        l1_two = args.level == 1 and not args.warm and (not hasattr(zz.lib, "zz_debug_l1_kernel") or zz.lib.zz_debug_l1_kernel(ctx._h) == 2)
        classic = os.environ.get("ZZFLATE_L1_KERNEL") == "classic"
        l1_kernel = ("k_encode_l1w" if classic else "k_encode_l1pw") if args.warm else ("k_encode_l1p" if l1_two else "k_encode_l1")
        l1_two = l1_two or (args.level == 1 and bool(args.warm) and not classic)
        scal = tj_scalar if (args.level == 1 and l1_two) or args.level in (2, 3) else None
        if insts and kms > 0 and args.level >= 1:
            l2_two = args.level in (2, 3) and not args.warm and os.environ.get("ZZFLATE_L2_KERNEL") != "classic"
            resident = 256 * (8 if args.level >= 4 else 9) * (3 if (l1_two or l2_two) else 2)     # k_encode_l1p / k_encode_l2p: two parsers + the emitter / helper per packet
            cpi, ghz = 5.0, 2.4
            bound_ms = insts * cpi / (resident * ghz * 1e9) * 1e3
            issue = {"bound": "issue", "instructions": insts, "cycles_per_instruction": cpi, "resident_wavefronts": resident,
                     "clock_ghz": ghz, "bound_ms": round(bound_ms, 3), "frac": round(bound_ms / kms, 4)}
        line = {
            "metric": "input GB/s compressed (whole node) + ratio, level 1, 1/2/4/8 MI355X",
            "value": round(total_n * args.steps / dt / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "repeats": regions["repeats"],
            "ms_per_step_min_max": [round(regions["ms_per_step_min"], 3), round(regions["ms_per_step_max"], 3)],
            "timed_seconds_total": round(regions["timed_seconds_total"], 3),
            "higher_is_better": True,
            "scaling": "weak",
            "backend": (("rccl" if backend == "nccl" else backend) + f" (torch.distributed {backend}, world size {dist.get_world_size()})") if multi else None,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "ratio": round(state["out_bytes"] / total_n, 4),
            "config": {
                "workload": f"{args.mib} MiB synthetic {args.gen} per GPU (zz_generate_device kind={args.gen}, seed "
                            f"{SEEDS[args.gen]:#x}), level {args.level}, {args.format} container, {P}-byte packets, "
                            f"input and output resident in HBM" + ((", nothing exchanged (--gather none)" if args.gather == "none" else ", shards gathered to rank 0 over RCCL" + (f" in {C} overlapped pieces" if C > 1 else ", each step's gather in flight under the next step's encoding")) if multi else ""),
                "level": args.level, "packet_size": P, "bytes_per_gpu": n, "format": args.format, "warm_window": args.warm,
            },
            "calls_in_flight": len(lanes) if (not multi) else None,
            "exchange": exchange if multi else None,
            "roofline": {
                "bound": "hbm", "kernel": ("k_l6_matches + k_encode_l2_t<32768, true>" if args.level >= 4 else
                                           ("k_encode_l2_t<32768, false>" if args.warm else "k_encode_l2_t<0, false>" if os.environ.get("ZZFLATE_L2_KERNEL") == "classic"
                                            else "k_encode_l2_t<0, false, true> (k_encode_l2p)") if args.level >= 2
                                           else l1_kernel if args.level == 1 else "k_encode_l0"),
                "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": round(kms, 4), "traffic": traffic,
                "traffic_source": traffic_src, "issue": issue, "scalar": scalar_bound(scal, kms),
            },
            "cpu_baseline": cpu,
            "check": check,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
