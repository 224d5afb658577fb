"""Diagnostic: dumps walk state of the first group of packet 0 (build with -DZZ_DEBUG_DUMP)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
lib = os.path.join(ROOT, "gpurun_out", "libzz_dbg.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DZZ_DEBUG_DUMP", "-o", lib,
                os.path.join(ROOT, "zzflate_amd/csrc/zz_api.hip"), os.path.join(ROOT, "zzflate_amd/csrc/zz_cxx_shim.cpp")], check=True)
L = ctypes.CDLL(lib)
u64, vp, ci, u32 = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32
h = vp(); assert L.zz_ctx_create(0, ctypes.byref(h)) == 0
L.zz_bound.restype = u64
data = bytes.fromhex(sys.argv[1]) if len(sys.argv) > 1 else b"\x15" * 5
n = len(data)
src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = L.zz_bound(u64(n), ci(2), ci(1), u32(32768))
dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
out = u64(0)
rc = L.zz_encode_device(h, vp(src.data_ptr()), u64(n), vp(dst.data_ptr()), u64(cap), ctypes.byref(out), ci(2), ci(1), u32(32768), vp(0))
prof = (ctypes.c_ulonglong * 16)()
L.zz_debug_read_prof(h, prof)
print("rc", rc, "out", dst[:out.value].cpu().numpy().tobytes().hex())
for nm, v in zip(["E", "lits", "mst", "usedB", "pos", "multimask"], prof[:6]):
    print(f"{nm:10s} {v:#x}")
for i in range(8):
    v = prof[8 + i]
    print(f"lane {i}: old {v >> 32} info {v & 0xffffffff:#x}")
