"""Multi-GPU join of a packet-sharded stream (replaces the reference's in-process
std::async fan-out + memmove join, zzflate.cpp:97-155, across devices).

Packets are independent and every non-final packet ends byte-aligned, so ranks own contiguous packet ranges
and the shard outputs concatenate by plain byte copy. The only exchange is
  1. an all-gather of (compressed bytes, checksum partial, input bytes) per rank, and
  2. ONE gather of the variable-size compressed shards onto the stream's root rank, as a grouped send/recv (RCCL has no
     gatherv); compressed bytes travel, never input bytes.
The root then adds the container header and the trailer from the combined checksums. The root is rank 0 unless the caller
names another: a job that produces stream after stream can rotate it (stream i onto rank i mod N), so that successive
streams do not all arrive over ONE GPU's seven inbound xGMI links while the other GPUs' inbound links idle.

Backend-agnostic: `nccl` (= RCCL over xGMI) with device tensors in bench.py, `gloo` with CPU tensors in the
tests. The per-shard encoder is whatever the caller ran (Context.encode_shard on a GPU).
"""
import torch

from . import Format, combine, crc32_combine, header, trailer


def shard_range(total_n, packet_size, rank, world):
    """Contiguous packet range of `rank`: returns (byte offset, byte count); boundaries are packet-aligned."""
    npk = (total_n + packet_size - 1) // packet_size
    per = (npk + world - 1) // world
    lo = min(rank * per * packet_size, total_n)
    hi = min((rank + 1) * per * packet_size, total_n)
    return lo, hi - lo


def combine_checksums(fmt, parts):
    """parts: [(checksum partial, input bytes)] in rank order -> the stream's Adler-32 / CRC-32 (or 0)."""
    fmt = int(fmt)
    if fmt == Format.Zlib:
        tot = 1                                   # adler32x(1, ...): zzflate.cpp:176
        for cks, n in parts:
            tot = combine(tot, cks, n)            # adler.cpp:5-15 semantics
        return tot
    if fmt == Format.Gzip:
        tot = 0
        for cks, n in parts:
            tot = crc32_combine(tot, cks, n)
        return tot
    return 0


class _Pending:
    """A gather whose transfers may still be in flight: `wait()` before touching `shard` or `out` again."""

    def __init__(self, reqs, total, keep):
        self.reqs, self.total, self.keep = reqs, total, keep

    def wait(self):
        for req in self.reqs:
            req.wait()
        self.reqs, self.keep = [], None
        return self.total


def gather_stream(dist, fmt, shard, shard_bytes, cks, n_in, out=None, group=None, wait=True, root=0):
    """Collective. `shard[:shard_bytes]` is this rank's compressed shard (uint8 tensor), `cks` its checksum
    partial, `n_in` its input byte count. Returns the total stream length (header + shards + trailer) on every rank -- it
    follows from the all-gathered sizes --; the bytes are written into `out` on rank `root` only (the other ranks pass
    out=None). With wait=False the grouped send/recv is left in flight and a handle comes back instead (`.wait()` -> the
    same result): the caller encodes its next stream into other buffers meanwhile, so the transfer over xGMI hides behind
    compute."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = shard.device
    meta = torch.tensor([shard_bytes, cks, n_in], dtype=torch.int64, device=dev)
    metas = torch.empty(3 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    m = metas.cpu().tolist()
    sizes, ckss, lens = m[0::3], m[1::3], m[2::3]
    head = header(fmt)
    ops, off = [], len(head)
    offs = []
    for r in range(world):
        offs.append(off)
        off += sizes[r]
    if rank == root:
        assert out is not None and out.numel() >= off + 8, "the root rank needs an output buffer"
        for r in range(world):
            if r != root and sizes[r]:
                ops.append(dist.P2POp(dist.irecv, out[offs[r]:offs[r] + sizes[r]], r, group))
        out[offs[root]:offs[root] + sizes[root]].copy_(shard[:sizes[root]])
    elif sizes[rank]:
        ops.append(dist.P2POp(dist.isend, shard[:sizes[rank]], root, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    tail = trailer(fmt, combine_checksums(fmt, list(zip(ckss, lens))), sum(lens))
    if rank == root:
        if head:
            out[:len(head)].copy_(torch.frombuffer(bytearray(head), dtype=torch.uint8))
        if tail:
            out[off:off + len(tail)].copy_(torch.frombuffer(bytearray(tail), dtype=torch.uint8))
    total = off + len(tail)
    pending = _Pending(reqs, total, (shard, out))
    return pending.wait() if wait else pending


class PipelinedGather:
    """The same join with the transfer hidden behind the encoding ("overlap by gathering ... as shards
    finish"). Every rank encodes its shard in a few packet-aligned pieces; after each piece the (bytes, checksum,
    input bytes) triples are all-gathered and the piece leaves for rank 0 as an asynchronous grouped send/recv
    while the next piece is being encoded. Pieces land in a per-rank staging region on rank 0 (their final offsets
    depend on sizes not known yet); `finish` waits for the transfers and assembles header + shards + trailer with
    device-to-device copies. All ranks must call `push` the same number of times per stream.
    """

    def __init__(self, dist, fmt, cap_per_rank, device, group=None):
        self.dist, self.fmt, self.group = dist, int(fmt), group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.cap = cap_per_rank
        self.device = device
        self.staging = torch.empty(cap_per_rank * self.world, dtype=torch.uint8, device=device) if self.rank == 0 else None
        self.begin()

    def begin(self):
        self.reqs = []
        self.fill = [0] * self.world          # bytes received so far per rank (rank 0's view; same on every rank)
        self.parts = [[] for _ in range(self.world)]   # per rank: [(checksum partial, input bytes)] in piece order
        self.keep = []                        # tensors that must outlive their sends

    def push(self, piece, nbytes, cks, n_in):
        dist, dev = self.dist, self.device
        meta = torch.tensor([nbytes, cks, n_in], dtype=torch.int64, device=dev)
        metas = torch.empty(3 * self.world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(metas, meta, group=self.group)
        m = metas.cpu().tolist()
        sizes, ckss, lens = m[0::3], m[1::3], m[2::3]
        ops = []
        for r in range(self.world):
            if sizes[r] > self.cap - self.fill[r]:
                raise RuntimeError("staging region too small")
            if self.rank == 0:
                dst = self.staging[r * self.cap + self.fill[r]: r * self.cap + self.fill[r] + sizes[r]]
                if r == 0:
                    dst.copy_(piece[:sizes[0]])
                elif sizes[r]:
                    ops.append(dist.P2POp(dist.irecv, dst, r, self.group))
            elif r == self.rank and sizes[r]:
                ops.append(dist.P2POp(dist.isend, piece[:sizes[r]], 0, self.group))
                self.keep.append(piece)
            self.fill[r] += sizes[r]
            self.parts[r].append((ckss[r], lens[r]))
        if ops:
            self.reqs.extend(dist.batch_isend_irecv(ops))

    def finish(self, out=None):
        for req in self.reqs:
            req.wait()
        self.keep = []
        if self.rank != 0:
            return None
        head = header(self.fmt)
        off = len(head)
        assert out is not None and out.numel() >= off + sum(self.fill) + 8
        for r in range(self.world):
            out[off: off + self.fill[r]].copy_(self.staging[r * self.cap: r * self.cap + self.fill[r]])
            off += self.fill[r]
        flat = [p for r in range(self.world) for p in self.parts[r]]
        tail = trailer(self.fmt, combine_checksums(self.fmt, flat), sum(n for _, n in flat))
        if head:
            out[:len(head)].copy_(torch.frombuffer(bytearray(head), dtype=torch.uint8))
        if tail:
            out[off: off + len(tail)].copy_(torch.frombuffer(bytearray(tail), dtype=torch.uint8))
        return off + len(tail)
