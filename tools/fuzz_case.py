"""Replays dumped fuzz cases (ZZ_FUZZ_DUMP=<n> python tools/fuzz_gpu.py ... writes gpurun_out/fuzz_case_<n>.in; copy them to
fuzz_cases/) in packet mode against the oracle, through the C ABI only, so that it also runs against an older build of the library:
    ZZFLATE_AMD_LIB=<lib.so> python tools/fuzz_case.py <case>:<level>:<format>:<packet>:<warm> ..."""
import ctypes, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
L = ctypes.CDLL(os.environ.get("ZZFLATE_AMD_LIB", os.path.join(ROOT, "zzflate_amd", "libzzflate_amd.so")))
O = ctypes.CDLL(os.path.join(ROOT, "oracle", "libzzoracle.so"))
u64, u32, ci, vp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p
L.zz_bound.restype = u64; L.zz_bound.argtypes = [u64, ci, ci, u32]
O.zzo_encode_packets_warm.restype = u64; O.zzo_encode_packets_warm.argtypes = [vp, u64, ctypes.c_char_p, u64, ci, ci, u64, u64]
h = vp(); assert L.zz_ctx_create(0, ctypes.byref(h)) == 0
for spec in sys.argv[1:]:
    c, lvl, fmt, P, warm = (int(x) for x in spec.split(":"))
    d = open(os.path.join(ROOT, "fuzz_cases", f"fuzz_case_{c}.in"), "rb").read()
    L.zz_ctx_set_warm_window(h, u32(warm)); L.zz_ctx_set_extended_levels(h, ci(1 if lvl > 3 else 0))
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = L.zz_bound(len(d), fmt, min(lvl, 3), P)
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    out = u64(0)
    rc = L.zz_encode_device(h, vp(src.data_ptr()), u64(len(d)), vp(dst.data_ptr()), u64(cap), ctypes.byref(out), ci(fmt), ci(lvl), u32(P), vp(0))
    got = dst[:out.value].cpu().numpy().tobytes() if rc == 0 else b""
    b = ctypes.create_string_buffer(2 * len(d) + 4096)
    wn = O.zzo_encode_packets_warm(b, len(b), d, len(d), fmt, lvl, P, warm if lvl < 4 else 0)
    want = b.raw[:wn]
    first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
    try:
        ok = zlib.decompressobj({0: 15, 1: 31, 2: -15}[fmt]).decompress(got) == d
    except zlib.error as e:
        ok = str(e)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"fuzz_case_{c}.got"), "wb").write(got)
    print(spec, "rc", rc, "n", len(d), "got", len(got), "want", len(want), "equal", got == want, "first diff", first, "inflates", ok, flush=True)
