"""The N > 1 path with the real shard encoder: two ranks share the one GPU of the test box, each encodes its packet
range with zz_encode_shard_device (left halo included), and zzflate_amd.sharded assembles the stream on rank 0 --
over gloo here (one GPU cannot host two RCCL ranks), over RCCL/xGMI on a multi-GPU node, same code. The result must
equal the single-call stream and the oracle's bit for bit. Also: `bench.py --gpus 2` launches its own ranks.
Needs a real MI355X: run with `-m gpu`."""
import json
import os
import socket
import subprocess
import time
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, zlib
ROOT = sys.argv[1]; fmt = int(sys.argv[2]); lvl = int(sys.argv[3]); P = int(sys.argv[4]); chunks = int(sys.argv[5]); out_path = sys.argv[6]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
import zzflate_amd as zz
from zzflate_amd import sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
ctx = zz.Context(0)
corpus = os.path.join(ROOT, "tests", "golden", "corpus")
data = b"".join(open(os.path.join(corpus, f), "rb").read() for f in ("lcet10.txt", "kennedy.xls", "ptt5", "alice29.txt")) * 2 + b"tail"
n_total = len(data)
whole = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
off, n = sharded.shard_range(n_total, P, rank, world)
halo = min(off, 65536)                       # the bytes in front of the shard that this rank holds too
cap = zz.bound(n, 2, lvl, P)
shard = torch.empty(cap, dtype=torch.uint8, device="cuda")
outbuf = torch.zeros(zz.bound(n_total, fmt, lvl, P) + 64, dtype=torch.uint8) if rank == 0 else None
if chunks <= 1:
    w, cks = ctx.encode_shard(whole.data_ptr() + off, n, shard, cap, halo=halo, is_last=(rank == world - 1), checksum=fmt, level=lvl, packet_size=P)
    assert ctx.verify_last() == (0, None)
    total = sharded.gather_stream(dist, fmt, shard[:w].cpu(), w, cks, n, outbuf)
else:
    pg = sharded.PipelinedGather(dist, fmt, cap + 4096, torch.device("cpu"))
    npk = (n + P - 1) // P
    per = (npk + chunks - 1) // chunks * P
    for c in range(chunks):
        a, b = min(c * per, n), min((c + 1) * per, n)
        w, cks = (0, 0)
        if b > a:
            w, cks = ctx.encode_shard(whole.data_ptr() + off + a, b - a, shard, cap, halo=halo + a, is_last=(rank == world - 1 and b == n),
                                      checksum=fmt, level=lvl, packet_size=P)
        pg.push(shard[:max(w, 1)].cpu(), w, cks, b - a)
    total = pg.finish(outbuf)
if rank == 0:
    from conftest import Oracle
    got = outbuf[:total].numpy().tobytes()
    cap1 = zz.bound(n_total, fmt, lvl, P)
    dst = torch.empty(cap1, dtype=torch.uint8, device="cuda")
    w1 = ctx.encode(whole, n_total, dst, cap1, fmt, lvl, P)
    single = dst[:w1].cpu().numpy().tobytes()
    want = Oracle().encode_packets(data, fmt, lvl, P)
    ok = got == single == want and zlib.decompressobj({0: 15, 1: 31, 2: -15}[fmt]).decompress(got) == data
    open(out_path, "w").write("ok" if ok else f"MISMATCH {len(got)} {len(single)} {len(want)}")
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(tmp_path, world, args):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    res = tmp_path / "result.txt"
    port = _free_port()
    procs, logs = [], []
    try:
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            logs.append(open(tmp_path / f"rank{r}.log", "w+"))
            procs.append(subprocess.Popen([sys.executable, str(script), ROOT] + [str(a) for a in args] + [str(res)], env=env,
                                          stdout=logs[-1], stderr=subprocess.STDOUT, text=True))
        # poll all ranks together: when one dies its peers would sit in a collective until the timeout, holding the GPU
        deadline = time.time() + 600
        failed = False
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
                failed = True
                break
            time.sleep(0.1)
        failed = failed or any(p.returncode != 0 for p in procs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
    outs = []
    for f in logs:
        f.seek(0)
        outs.append(f.read())
        f.close()
    assert not failed, "\n".join(outs)
    return res.read_text()


@pytest.mark.parametrize("fmt,lvl,P,chunks", [(0, 1, 32768, 1), (1, 2, 32768, 1), (0, 3, 32768, 3), (2, 0, 4096, 1), (1, 1, 8192, 2)])
def test_two_ranks_encode_shards_and_gather(tmp_path, fmt, lvl, P, chunks):
    assert _run_ranks(tmp_path, 2, [fmt, lvl, P, chunks]) == "ok"


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself and reports n_gpus 2; a world size
    that does not match --gpus is an error, not a silent single-GPU run."""
    env = dict(os.environ, ZZ_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mib", "64",
                        "--no-cpu", "--no-extra"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "gloo" in line["backend"] and "world size 2" in line["backend"]
    assert line["check"]["device_inflate"]["bad"] == 0 and line["check"]["inflate_prefix_ok"]
    assert line["check"]["device_inflate"]["packets"] == 2 * (64 << 20) // 32768
    # what a scaling run needs to tell encoder scaling from link saturation
    ex = line["exchange"]
    assert ex["encode_only"]["value"] > 0 and ex["encode_only"]["ms_per_step"] > 0 and ex["gather_ms_serial"] > 0
    hl, tl = 2, 4
    total = round(line["ratio"] * 2 * (64 << 20))
    assert 0 < ex["bytes_into_rank0_per_step"] < total - hl - tl          # rank 1's shard, and only that
    assert abs(ex["bytes_into_rank0_per_step"] - (total - hl - tl) / 2) < 0.2 * total
    # --gather none: the encoders alone
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mib", "64",
                         "--no-cpu", "--no-extra", "--gather", "none"], env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    l2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert l2["n_gpus"] == 2 and l2["exchange"] is None and "nothing exchanged" in l2["config"]["workload"]
    assert l2["check"]["device_inflate"]["bad"] == 0 and l2["check"]["inflate_prefix_ok"]
    # --gather-root rotate: step i's stream lands on rank i mod 2 (three steps: both ranks take a turn); the line, the
    # device inflate of every packet and the zlib check of the stream on rank 0 must come out the same
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--mib", "64",
                         "--no-cpu", "--no-extra", "--gather-root", "rotate"], env=env, capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    l3 = json.loads([l for l in r3.stdout.splitlines() if l.startswith("{")][-1])
    assert l3["n_gpus"] == 2 and l3["ratio"] == line["ratio"] and l3["exchange"]["gather_root"] == "rotate"
    assert l3["exchange"]["gather_ms_serial_onto_last_rank"] > 0
    assert l3["check"]["device_inflate"]["bad"] == 0 and l3["check"]["inflate_prefix_ok"]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"], env=dict(env, WORLD_SIZE="1"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_rccl_path_with_one_rank(tmp_path):
    """The distributed path of bench.py on RCCL with the one rank a one-GPU box can host: communicator creation, the
    size/checksum all-gather, the all-reduce of the step time, barriers and the assembly on rank 0 run over RCCL (backend
    "nccl" on ROCm); only the send/recv of the gather itself needs a second GPU (covered over gloo above)."""
    env = dict(os.environ, ZZ_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ZZ_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--mib", "64",
                        "--no-cpu", "--no-extra"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["backend"].startswith("rccl") and "world size 1" in line["backend"]
    assert line["check"]["device_inflate"]["bad"] == 0 and line["check"]["inflate_prefix_ok"]
    assert line["exchange"]["bytes_into_rank0_per_step"] == 0 and line["exchange"]["encode_only"]["value"] > 0
