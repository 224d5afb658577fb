"""What the occupancy arithmetic of DESIGN.md 4 rests on, checked against the compiler (CPU only: hipcc cross-compiles gfx950):
LDS per workgroup small enough for nine workgroups per CU, registers small enough for the wavefronts those need, no scratch in
the level-1 kernels, and scratch in the level-2 kernel touched at kernel entry and once per packet only -- never inside the block
loops (tools/scratch_report.py is the long form; profiles/r03_resource_usage.txt its output)."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def report():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scratch_report.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def usage(report, mangled):
    m = re.search(r"Function Name: %s\n(.*?)\n\n" % re.escape(mangled), report, flags=re.S)
    assert m, mangled
    return {k.strip(): v.strip() for k, v in (ln.split(":", 1) for ln in m.group(1).splitlines() if ":" in ln)}


def test_level1_kernels_fit_nine_workgroups_per_cu_without_scratch(report):
    for name in ("_ZN2zz11k_encode_l1E16zz_packet_params", "_ZN2zz12k_encode_l1wE16zz_packet_params"):
        u = usage(report, name)
        assert int(u["LDS Size [bytes/block]"]) * 9 <= 160 * 1024          # nine 16 KiB hash tables (+ ring, + slots) per CU
        assert int(u["VGPRs"]) <= 96 and int(u["ScratchSize [bytes/lane]"]) == 0
    assert re.search(r"k_encode_l1E16zz_packet_params: 0 scratch_store, 0 scratch_load", report)


def test_two_parser_level1_kernel_keeps_nine_workgroups_of_three_wavefronts(report):
    """k_encode_l1p (the headline kernel): three wavefronts per packet, nine packets per CU = 27 wavefronts, seven on one SIMD at
    worst -- 512 / 7 = 73 registers, so at most 72 VGPRs; LDS in nine shares of the CU's 160 KiB at the allocation granule; no
    scratch. 44 bytes of LDS are what stands between nine workgroups and eight."""
    u = usage(report, "_ZN2zz12k_encode_l1pE16zz_packet_params")
    lds = int(u["LDS Size [bytes/block]"])
    assert -(-lds // 512) * 512 * 9 <= 160 * 1024, lds                    # allocated in 512-byte granules (17,920 = 35 of them)
    assert int(u["VGPRs"]) <= 72 and int(u["ScratchSize [bytes/lane]"]) == 0 and int(u["VGPRs Spill"]) == 0
    assert re.search(r"k_encode_l1pE16zz_packet_params: 0 scratch_store, 0 scratch_load", report)


def test_level2_kernel_scratch_stays_out_of_the_block_loops(report):
    u = usage(report, "_ZN2zz13k_encode_l2_tILj0ELb0ELb0EEEvNS_12zz_l2_paramsE")
    assert int(u["LDS Size [bytes/block]"]) * 9 <= 160 * 1024 and int(u["VGPRs"]) <= 96
    m = re.search(r"k_encode_l2_tILj0ELb0ELb0EEEvNS_12zz_l2_paramsE: (\d+) scratch_store, (\d+) scratch_load instructions\n((?:  depth .*\n)*)", report)
    assert m
    # round 4: no scratch at all. Round 5 (this is the fallback kernel now; the code-length run lengths became a wave-wide step):
    # two lane constants are stored at kernel entry and read back on the backward extension's byte-wise edge path. Nothing may be
    # STORED in the block loops (depth >= 2), and the loads there stay a handful.
    stores = [int(d) for d, kind in re.findall(r"depth (\d+):\s+\d+ scratch_(store|load)", m.group(3)) if kind == "store"]
    deep_loads = sum(int(c) for d, c, kind in re.findall(r"depth (\d+):\s+(\d+) scratch_(store|load)", m.group(3)) if kind == "load" and int(d) >= 2)
    assert (not stores or max(stores) <= 1) and deep_loads <= 4 and int(u["ScratchSize [bytes/lane]"]) <= 16, m.group(0)


def test_two_parser_level2_kernel_fits_nine_workgroups_of_three_wavefronts(report):
    """k_encode_l2_t<0, false, true> ("k_encode_l2p": levels 2,3 on two parsing wavefronts + the helper): 27 wavefronts per CU want at
    most 72 VGPRs and nine LDS shares. The compiler is held to 72 (launch bounds) and spills what does not fit: values that are
    invariant over the packets, stored once at kernel entry and read back per packet or on the out-of-line path of the walk (a
    "16 or more" / "8 or more" length being extended) -- never stored inside the block loops."""
    name = "_ZN2zz13k_encode_l2_tILj0ELb0ELb1EEEvNS_12zz_l2_paramsE"
    u = usage(report, name)
    lds = int(u["LDS Size [bytes/block]"])
    assert -(-lds // 512) * 512 * 9 <= 160 * 1024, lds
    assert int(u["VGPRs"]) <= 72 and int(u["ScratchSize [bytes/lane]"]) <= 160
    m = re.search(r"k_encode_l2_tILj0ELb0ELb1EEEvNS_12zz_l2_paramsE: (\d+) scratch_store, (\d+) scratch_load instructions\n((?:  depth .*\n)*)", report)
    assert m
    stores = [int(d) for d, kind in re.findall(r"depth (\d+):\s+\d+ scratch_(store|load)", m.group(3)) if kind == "store"]
    assert not stores or max(stores) <= 1, m.group(0)


def test_extended_level_kernels(report):
    """k_l6_matches owns a CU: its LDS (sorted array + counters + ring) must fit the 160 KiB a workgroup can have, its 16
    wavefronts need <= 128 VGPRs, and it must not spill. The encode kernel behind it has no hash table: eleven workgroups per CU
    (LDS), i.e. six wavefronts on two of the SIMDs: at most 80 VGPRs."""
    for depth in (2, 4, 8):
        u = usage(report, "_ZN2zz12k_l6_matchesILi%dEEEvNS_13zz_l6m_paramsE" % depth)
        assert 128 * 1024 < int(u["LDS Size [bytes/block]"]) <= 160 * 1024
        assert int(u["VGPRs"]) <= 128 and int(u["ScratchSize [bytes/lane]"]) == 0 and int(u["VGPRs Spill"]) == 0
    u = usage(report, "_ZN2zz13k_encode_l2_tILj32768ELb1ELb0EEEvNS_12zz_l2_paramsE")
    lds = -(-int(u["LDS Size [bytes/block]"]) // 512) * 512                      # allocated in 512-byte granules
    assert lds * 11 <= 160 * 1024 < lds * 12 and int(u["VGPRs"]) <= 80
