"""Host-buffer entry point (zz_encode): PCIe-inclusive rate, for DESIGN.md (never the bench `value`).
Times the C ABI call alone (buffers allocated and touched beforehand), then the Python mirror with its copies."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zzflate_amd as zz
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
data = np.frombuffer(zz.generate_host(zz.GEN_TEXT, 0x5EED0002, 0, n), dtype=np.uint8)
L = zz.lib
for lvl in (1, 2):
    cap = zz.bound(n, 0, lvl, 32768)
    dst = np.zeros(cap, dtype=np.uint8)
    cfg = zz._cfg(zz.Config(zz.Format.Zlib, lvl, True))
    best = 1e9
    for it in range(3):
        ln = ctypes.c_uint64(cap)
        t = time.perf_counter()
        rc = L.zz_encode(dst.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ln), data.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(cfg))
        dt = time.perf_counter() - t
        assert rc == 0, L.zz_last_error()
        best = min(best, dt)
    print(f"level {lvl}: zz_encode host->host {n / best / 1e9:.2f} GB/s (H2D + encode + D2H pipelined in slabs), ratio {ln.value / n:.4f}")
small = bytes(data[:256 << 20])
t = time.perf_counter(); out = zz.ZzFlateEncode(small, zz.Config(zz.Format.Zlib, 1, True)); dt = time.perf_counter() - t
print(f"python mirror ZzFlateEncode (256 MiB, with its bytes() and buffer copies): {len(small) / dt / 1e9:.2f} GB/s")
