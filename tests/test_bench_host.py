"""Host-side pieces of bench.py that need no GPU: the core count the CPU baseline may use, and the native timing drivers
(oracle/ref_harness.cpp zzref_bench, oracle/zzoracle.c zzo_bench) on a small sample."""
import ctypes
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402
import zzflate_amd as zz  # noqa: E402


def test_effective_cores_is_affinity_cut_down_by_the_quota():
    cores, affinity, quota = bench.effective_cores()
    assert 1 <= cores <= affinity
    if quota is not None:
        assert cores <= max(1, int(quota + 0.5))


def _drive(lib, fn_name, sample, mode, threads, level):
    L = ctypes.CDLL(lib)
    fn = getattr(L, fn_name)
    u64, dbl = ctypes.c_uint64, ctypes.c_double
    fn.restype = dbl
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, u64, u64, u64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                   ctypes.POINTER(u64), ctypes.POINTER(dbl)]
    buf = ctypes.create_string_buffer(sample, len(sample) + 64)
    nsl, slb = 4, len(sample) // 4
    produced, secs = (u64 * nsl)(), (dbl * threads)()
    wall = fn(mode, ctypes.addressof(buf), nsl, slb, 2 * nsl, threads, 2, level, 32768, produced, secs)
    return wall, list(produced), list(secs)


def test_native_timing_drivers_do_the_work_they_time(oracle):
    sample = zz.generate_host(zz.GEN_TEXT, 0x5EED0002, 0, 4 << 20)
    want_packets = []
    for s in range(4):
        sl = sample[s << 20:(s + 1) << 20]
        # packet mode over a slice of the sample: the oracle's packets, none of them final except the sample's last
        tot = 0
        for off in range(0, 1 << 20, 32768):
            tot += len(oracle.packet(sample, 1, (s << 20) + off, 32768, s == 3 and off + 32768 == 1 << 20))
        want_packets.append(tot)
    libs = [(os.path.join(ROOT, "oracle", "libzzoracle.so"), "zzo_bench")]
    ref = os.path.join(ROOT, "oracle", "_ref", "libzzref.so")
    if os.path.exists(ref):
        libs.append((ref, "zzref_bench"))
    for lib, name in libs:
        for threads in (1, 3):
            wall, produced, secs = _drive(lib, name, sample, 1, threads, 1)
            assert wall > 0 and all(x > 0 for x in secs) and max(secs) <= wall * 1.5 + 0.05
            assert produced == want_packets, (name, threads)
            wall, produced, secs = _drive(lib, name, sample, 0, threads, 1)
            assert all(0 < p < (1 << 20) for p in produced)      # text compresses; one whole-slice stream per slice


def test_profile_summaries_keep_template_instantiations_apart():
    """tools/summarize_pmc.py: `k_encode_l2_t<0u, false>` (level 2) and `k_encode_l2_t<32768u, true>` (the extended levels'
    encode kernel) are different kernels in profiles/traffic.json; return type and parameter list go, template arguments stay."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("summarize_pmc", os.path.join(ROOT, "tools", "summarize_pmc.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.short('"void zz::k_encode_l2_t<0u, false>(zz::zz_l2_params)"') == "zz::k_encode_l2_t<0u, false>"
    assert m.short('"void zz::k_encode_l2_t<32768u, true>(zz::zz_l2_params)"') == "zz::k_encode_l2_t<32768u, true>"
    assert m.short('"zz::k_encode_l1(zz_packet_params)"') == "zz::k_encode_l1"
    assert m.short('"void zz::k_l6_matches<8>(zz::zz_l6m_params)"') == "zz::k_l6_matches<8>"
    assert m.short("k_generate(int, unsigned long, unsigned long, unsigned char*, unsigned long)") == "k_generate"
