"""Level-1 throughput per family of the twelve-family mix (16 MiB segments of `--gen mix`), one segment tiled to 256 MiB.
Run once per kernel:  ZZFLATE_L1_KERNEL=classic python3 tools/family_rates.py ; python3 tools/family_rates.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zzflate_amd as zz
names = ["prose", "XML", "C source", "HTML", "binary records", "bi-level image", "16-bit gradients", "database dump", "executable", "random", "DNA-like", "log text"]
ctx = zz.Context(0)
seg = 16 << 20
mix = torch.empty(12 * seg + 64, dtype=torch.uint8, device="cuda")
ctx.generate(zz.GEN_MIX, 0x5EED0004, 0, mix, 12 * seg)
n = 256 << 20
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cap = zz.bound(n, 0, level, 32768)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
print("kernel:", os.environ.get("ZZFLATE_L1_KERNEL", "default"))
for f, nm in enumerate(names):
    src = mix[f * seg:(f + 1) * seg].repeat(n // seg)
    w = ctx.encode(src, n, dst, cap, 0, level)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): w = ctx.encode(src, n, dst, cap, 0, level)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"  {nm:18s} {n / dt / 1e9:7.1f} GB/s  ratio {w / n:.4f}", flush=True)
