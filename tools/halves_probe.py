"""Probe (manual): one call's work as two half-shards on two contexts and two streams, against one call -- what an in-call split
of the encode kernel could gain (the second half's kernel fills the CUs the first half's last packets leave idle and runs under
its scan / compaction)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zzflate_amd as zz
n = 1 << 30
P = 32768
ca, cb, c1 = zz.Context(0), zz.Context(0), zz.Context(0)
src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
c1.generate(0, 0x5EED0002, 0, src, n)
cap = zz.bound(n, 2, 1, P)
d1 = torch.empty(cap, dtype=torch.uint8, device="cuda")
da = torch.empty(cap // 2 + 4096, dtype=torch.uint8, device="cuda")
db = torch.empty(cap // 2 + 4096, dtype=torch.uint8, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for frac in (0.5, 0.6, 0.7):
    h = int(n * frac) // P * P
    def pair():
        ca.encode_shard_async(src, h, da, da.numel(), halo=0, is_last=False, checksum=0, level=1, packet_size=P, stream=sa.cuda_stream)
        cb.encode_shard_async(src[h:], n - h, db, db.numel(), halo=h, is_last=True, checksum=0, level=1, packet_size=P, stream=sb.cuda_stream)
        wa, _ = ca.finish_shard(0); wb, _ = cb.finish_shard(0)
        return wa + wb
    def single():
        return c1.encode_shard(src, n, d1, cap, halo=0, is_last=True, checksum=0, level=1, packet_size=P)[0]
    for name, fn in (("single", single), ("halves %.1f" % frac, pair), ("single", single), ("halves %.1f" % frac, pair)):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): w = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f"{name:12s} {n / dt / 1e9:7.2f} GB/s  {dt * 1e3:.3f} ms  bytes {w}", flush=True)
