"""Source-level drop-in: a C++ caller written against the reference's header (include/zzflate.h mirrors
zzflate/zzflate.h:8-19) compiles with g++ -std=c++14 and links against libzzflate_amd.so unchanged."""
import os
import subprocess

import pytest

import zzflate_amd as zz
from conftest import ROOT


def build_caller(tmp_path):
    exe = str(tmp_path / "dropin_caller")
    libdir = os.path.dirname(zz._build.LIB)
    subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cxx", "dropin_caller.cpp"), "-o", exe, "-L", libdir, "-lzzflate_amd",
                    "-lz", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_links_and_reports_errors_without_gpu(tmp_path):
    exe = build_caller(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "corpus", "alice29.txt")], capture_output=True, text=True)
    lines = r.stdout.split("\n")
    assert "OK combine" in lines and "OK crc" in lines          # host utilities always work
    try:
        import torch
        gpu = torch.cuda.is_available()
    except ImportError:
        gpu = False
    if not gpu:
        assert "ERR encode" in lines and "ERR callback" in lines   # no device => error convention, no CPU fallback
        assert "ERR roundtrip 1" in lines and "ERR gzip" in lines


@pytest.mark.gpu
def test_cxx_caller_round_trips_on_gpu(tmp_path):
    exe = build_caller(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "corpus", "alice29.txt")], capture_output=True, text=True)
    lines = [l for l in r.stdout.split("\n") if l]
    assert r.returncode == 0, r.stdout + r.stderr
    assert lines[0] == "OK 92346" and lines[1].startswith("OK 65734 ")   # SURVEY App. D packet-mode sizes
    # the reference's own callers (zztest/Test.cpp:202-282): threaded=false, level 1 into max(200, n) bytes, levels 2,3
    # through the callback, gzip level 1 into n bytes. Levels 2,3 are App. D's whole-stream sizes.
    assert lines[2].startswith("OK roundtrip 1 ") and lines[3] == "OK roundtrip 2 64090" and lines[4] == "OK roundtrip 3 64090"
    assert lines[5].startswith("OK gzip ")
