"""What would a call cut into packet ranges on two streams bring? (the joins' 4.5 %.) A probe through the existing
shard entry points, no library change: 1 GiB of text at level 1 as ONE call, and as K shards (contiguous packet ranges, each with
its own scan + compaction + checksum fold) enqueued alternately on two contexts / two streams, the first stream at high priority
or not. The shards' bytes land in separate buffers: the concatenation a real call would add is a copy the last shard alone
exposes, so this is the idea's upper bound.   python tools/split_call_probe.py [level] [mib]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zzflate_amd as zz
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
P = 32768
dev = torch.device("cuda:0")
src = torch.empty(n + 64, dtype=torch.uint8, device=dev)
c0 = zz.Context(0)
c0.generate(zz.GEN_TEXT, 0x5EED0002, 0, src, n)
cap = zz.bound(n, 0, lvl, P)
dst = torch.empty(cap, dtype=torch.uint8, device=dev)
R = 10


def timed(f):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(R): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / R


one = timed(lambda: c0.encode(src, n, dst, cap, 0, lvl))
print(f"one call: {one * 1e3:.3f} ms = {n / one / 1e9:.2f} GB/s")
ctxs = [zz.Context(0), zz.Context(0)]
for prio in (False, True):
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    streams = [torch.cuda.Stream(device=dev, priority=(-1 if prio else 0)), torch.cuda.Stream(device=dev, priority=0)]
    for K in (2, 4, 8):
        npk = n // P
        cuts = [(npk * i // K) * P for i in range(K + 1)]
        dsts = [torch.empty(zz.bound(cuts[i + 1] - cuts[i], 0, lvl, P), dtype=torch.uint8, device=dev) for i in range(K)]

        def run():
            total = 0
            for i in range(K):
                c = ctxs[i & 1]
                if i >= 2: total += c.finish_shard()[0]            # this context's shard of two turns ago
                a, b = cuts[i], cuts[i + 1]
                c.encode_shard_async(src.data_ptr() + a, b - a, dsts[i], dsts[i].numel(), halo=a, is_last=(i == K - 1), level=lvl,
                                     packet_size=P, stream=streams[i & 1].cuda_stream)
            for i in range(max(0, K - 2), K): total += ctxs[i & 1].finish_shard()[0]
            return total
        t = timed(run)
        print(f"{K} shards on two streams{' (first stream high priority)' if prio else ''}: {t * 1e3:.3f} ms = {n / t / 1e9:.2f} GB/s "
              f"({(one / t - 1) * 100:+.1f} % against one call), bytes {run()}")
