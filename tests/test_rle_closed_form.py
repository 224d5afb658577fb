"""CPU check of the closed forms behind the wave-wide run-length step of levels 2,3 (zzflate_amd/csrc/zz_level2.h, rle_lengths_w):
what a run of c equal code lengths v becomes. The loops restated here are the reference's AddRecords (huffman.cpp:191-216) as the
oracle and the one-lane kernel path (rle_add) have them; the closed form is what a lane of the kernel computes for its run."""
import pytest


def add_records_loop(value, count):
    """the reference's loops: list of (symbol, payload) records for `count` lengths `value`"""
    out = []
    if count == 0:
        return out
    if value == 0:
        while count >= 3:
            w = min(count, 138)
            count -= w
            out.append((17 if w < 11 else 18, w))
    else:
        out.append((value, 0))
        count -= 1
        while count >= 3:
            w = min(count, 6)
            count -= w
            out.append((16, w))
    out += [(value, 0)] * count
    return out


def add_records_closed_form(value, c):
    """what rle_lengths_w's lane writes for a run: q full records, then one for a remainder >= 3, else the remainder spelled out"""
    out = []
    if value == 0:
        q, r = divmod(c, 138)
        out += [(18, 138)] * q
        out += [(17 if r < 11 else 18, r)] if r >= 3 else [(0, 0)] * r
    else:
        q, r = divmod(c - 1, 6)
        out.append((value, 0))
        out += [(16, 6)] * q
        out += [(16, r)] if r >= 3 else [(value, 0)] * r
    return out


def test_every_run_length_of_an_alphabet():
    for value in (0, 1, 7, 15):
        for c in range(1, 320):
            assert add_records_closed_form(value, c) == add_records_loop(value, c), (value, c)


def test_record_count_and_meta_frequencies():
    """the prefix sum places a run's records by their number; the meta frequencies take one add per kind"""
    for value in (0, 3):
        for c in range(1, 320):
            recs = add_records_loop(value, c)
            if value == 0:
                q, r = divmod(c, 138)
                assert len(recs) == q + (1 if r >= 3 else r)
                assert sum(1 for s, _ in recs if s == 18) == q + (1 if r >= 11 else 0)
                assert sum(1 for s, _ in recs if s == 17) == (1 if 3 <= r < 11 else 0)
                assert sum(1 for s, _ in recs if s == 0) == (r if r < 3 else 0)
            else:
                q, r = divmod(c - 1, 6)
                assert len(recs) == 1 + q + (1 if r >= 3 else r)
                assert sum(1 for s, _ in recs if s == 16) == q + (1 if r >= 3 else 0)
                assert sum(1 for s, _ in recs if s == value) == 1 + (r if r < 3 else 0)
