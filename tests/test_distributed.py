"""The N > 1 path on CPU: world_size-2 gloo run of zzflate_amd.sharded (shard split, size/checksum
all-gather, grouped send/recv gather, container assembly). No GPU here, so each rank's shard bytes come
from the oracle's packet function standing in for Context.encode_shard -- what is under test is the
exchange and assembly, whose result must equal the single-call packet stream bit for bit."""
import os
import socket
import sys
import zlib

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


def _worker(rank, world, port, fname, fmt, lvl, P, result_path, chunks=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import Oracle
    import zzflate_amd as zz
    from zzflate_amd import sharded
    oracle = Oracle()
    data = open(fname, "rb").read()
    off, n = sharded.shard_range(len(data), P, rank, world)
    # stand-in shard encoder: the oracle's packets over this rank's packet range (whole input visible, as
    # the halo contract of zz_encode_shard_device requires)
    npk_total = (len(data) + P - 1) // P
    out = b""
    for k in range(off // P, (off + n + P - 1) // P):
        ln = min(P, len(data) - k * P)
        out += oracle.packet(data, lvl, k * P, ln, k == npk_total - 1)
    part = data[off:off + n]
    cks = zz.adler32x(0, part) if fmt == 0 else (zz.crc32(part) if fmt == 1 else 0)
    outbuf = torch.zeros(2 * len(data) + 4096, dtype=torch.uint8) if rank == 0 else None
    if chunks <= 1:
        shard = torch.frombuffer(bytearray(out) if out else bytearray(1), dtype=torch.uint8)
        # two streams in flight at once (what bench.py does across steps), then the plain blocking form
        outbuf2 = torch.zeros_like(outbuf) if rank == 0 else None
        h1 = sharded.gather_stream(dist, fmt, shard, len(out), cks, n, outbuf2, wait=False)
        h2 = sharded.gather_stream(dist, fmt, shard.clone(), len(out), cks, n, outbuf, wait=False)
        t1, total = h1.wait(), h2.wait()
        if rank == 0:
            assert t1 == total and bytes(outbuf2[:t1].numpy()) == bytes(outbuf[:total].numpy())
        assert sharded.gather_stream(dist, fmt, shard, len(out), cks, n, outbuf) == total
        # a rotating gather root (bench.py --gather-root rotate: stream i onto rank i mod N): the same stream assembled on rank 1
        outbuf_r = torch.zeros(2 * len(data) + 4096, dtype=torch.uint8) if rank == 1 else None
        tr = sharded.gather_stream(dist, fmt, shard, len(out), cks, n, outbuf_r, root=1)
        rot_ok = torch.tensor([1 if (rank != 1 or (tr == total and outbuf_r[:tr].numpy().tobytes() == oracle.encode_packets(data, fmt, lvl, P))) else 0])
        dist.all_reduce(rot_ok, op=dist.ReduceOp.MIN)
        assert int(rot_ok.item()) == 1 and tr == total
    else:
        # pipelined variant: the shard leaves in `chunks` packet-aligned pieces
        pg = sharded.PipelinedGather(dist, fmt, 2 * len(data) + 4096, torch.device("cpu"))
        pk_lo, pk_hi = off // P, (off + n + P - 1) // P
        per = (pk_hi - pk_lo + chunks - 1) // chunks
        for c in range(chunks):
            a, b = min(pk_lo + c * per, pk_hi), min(pk_lo + (c + 1) * per, pk_hi)
            piece = b"".join(oracle.packet(data, lvl, k * P, min(P, len(data) - k * P), k == npk_total - 1) for k in range(a, b))
            pin = data[a * P: min(b * P, len(data))] if b > a else b""
            pc = zz.adler32x(0, pin) if fmt == 0 else (zz.crc32(pin) if fmt == 1 else 0)
            pg.push(torch.frombuffer(bytearray(piece) if piece else bytearray(1), dtype=torch.uint8), len(piece), pc, len(pin))
        total = pg.finish(outbuf)
    if rank == 0:
        got = outbuf[:total].numpy().tobytes()
        want = oracle.encode_packets(data, fmt, lvl, P)
        ok = got == want and zlib.decompressobj({0: 15, 1: 31, 2: -15}[fmt]).decompress(got) == data
        open(result_path, "w").write("ok" if ok else f"MISMATCH {len(got)} {len(want)}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fmt,lvl,P", [(0, 1, 32768), (1, 2, 32768), (2, 0, 4096), (0, 2, 32768)])
def test_two_rank_gather_equals_single_stream(tmp_path, fmt, lvl, P):
    fname = os.path.join(ROOT, "tests", "golden", "corpus", "alice29.txt")
    res = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), fname, fmt, lvl, P, res), nprocs=2, join=True)
    assert open(res).read() == "ok"


@pytest.mark.parametrize("fmt,lvl,P,chunks", [(0, 1, 4096, 4), (1, 2, 8192, 3), (0, 2, 32768, 2)])
def test_two_rank_pipelined_gather(tmp_path, fmt, lvl, P, chunks):
    """The overlapped exchange (pieces sent while the next piece is encoded) assembles the same stream."""
    fname = os.path.join(ROOT, "tests", "golden", "corpus", "lcet10.txt")
    res = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), fname, fmt, lvl, P, res, chunks), nprocs=2, join=True)
    assert open(res).read() == "ok"


def test_shard_ranges_cover_input_and_are_packet_aligned():
    sys.path.insert(0, ROOT)
    from zzflate_amd import sharded
    for total in (0, 1, 32768, 32769, 10 * 32768 + 5, 1 << 30):
        for world in (1, 2, 4, 8):
            pos = 0
            for r in range(world):
                off, n = sharded.shard_range(total, 32768, r, world)
                assert off == pos and off % 32768 == 0 or n == 0
                pos += n
            assert pos == total


def test_checksum_combine_over_ranks():
    sys.path.insert(0, ROOT)
    import zzflate_amd as zz
    from zzflate_amd import sharded
    import random
    rng = random.Random(9)
    data = bytes(rng.getrandbits(8) for _ in range(200000))
    cuts = [0, 65536, 131072, 200000]
    parts = [data[a:b] for a, b in zip(cuts, cuts[1:])]
    assert sharded.combine_checksums(0, [(zz.adler32x(0, p), len(p)) for p in parts]) == zlib.adler32(data)
    assert sharded.combine_checksums(1, [(zz.crc32(p), len(p)) for p in parts]) == zlib.crc32(data)
