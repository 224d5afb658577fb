"""Host-buffer entry point (zz_encode): PCIe-inclusive rate, for DESIGN.md (never the bench `value`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zzflate_amd as zz
n = 256 << 20
data = zz.generate_host(zz.GEN_TEXT, 0x5EED0002, 0, n)
for lvl in (1, 2):
    cfg = zz.Config(zz.Format.Zlib, lvl, True)
    zz.ZzFlateEncode(data[:1 << 20], cfg)
    t = time.perf_counter(); out = zz.ZzFlateEncode(data, cfg); dt = time.perf_counter() - t
    t = time.perf_counter(); out = zz.ZzFlateEncode(data, cfg); dt = min(dt, time.perf_counter() - t)
    print(f"level {lvl}: host->host {n / dt / 1e9:.2f} GB/s incl. H2D + D2H + ctypes copies, ratio {len(out) / n:.4f}")
