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


def test_timed_regions_are_repeated_to_a_second_and_summarised_by_their_median():
    """bench.py times the contract's region (exactly --steps steps) again and again until the regions add up to a second; the
    line's ms_per_step and value come from the median region, min and max ride along."""
    times = []
    while bench.want_another_region(times):
        times.append(0.153)                      # the driver's --steps 20 at 7.65 ms per step
    assert len(times) == 7 and sum(times) >= 1.0
    assert not bench.want_another_region([1.2]) and bench.want_another_region([])
    assert not bench.want_another_region([0.001] * 64)           # bounded
    assert bench.want_another_region([], 0.0) and not bench.want_another_region([0.01], 0.0)     # --min-seconds 0: exactly one region
    r = bench.summarize_regions([0.160, 0.150, 0.155, 0.152, 0.300], 20)
    assert r["repeats"] == 5 and abs(r["ms_per_step"] - 7.75) < 1e-9
    assert abs(r["ms_per_step_min"] - 7.5) < 1e-9 and abs(r["ms_per_step_max"] - 15.0) < 1e-9
    r = bench.summarize_regions([0.2, 0.1], 10)
    assert abs(r["ms_per_step"] - 15.0) < 1e-9 and r["repeats"] == 2


def test_scalar_unit_bound_arithmetic():
    """roofline.scalar: (SALU + branch) / (256 CUs x 2.4 GHz x kernel time) per CU per cycle against the measured 0.92"""
    b = bench.scalar_bound(2_660_000_000, 7.34)
    assert abs(b["per_cu_per_cycle"] - 2.66e9 / (256 * 2.4e9 * 7.34e-3)) < 1e-4 and 0.6 < b["frac"] < 0.7
    assert bench.scalar_bound(None, 7.0) is None and bench.scalar_bound(1, 0) is None
