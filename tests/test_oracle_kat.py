"""Known-answer tests of the CPU oracle (SURVEY.md §8c list), CPU only.  The oracle is "parity unpinned" by the
reference (it ships no tests); these analytic cases, each derived from the cited reference lines, are what pins it."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import kat_cases as kc

F = np.float32


def make(cfg, params):
    return orc.OracleMap(dict(cfg, bin_order=1), params, np.zeros(1000000, np.float32))


@pytest.mark.parametrize("case", kc.ALL_CASES, ids=lambda c: c.__name__)
def test_case(case):
    case(make)


def test_weight_closed_form():
    m = make(kc.K0, kc.PARAMS)
    kc.case_weight_closed_form(make, m.pdf_table())


def test_index_round_trip_all_octants():
    """(1) pos -> voxel -> min corner (operations.h:864-900, 970-983) incl. the (-1,0) truncation case."""
    m = make(kc.K0, kc.PARAMS)
    size = 0.5
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                p = (sx * 1.3, sy * 2.2, sz * 3.4)
                v = m.pos_to_voxel(*p)
                assert v == kc.voxel_of(kc.K0, *p)
                c = m.voxel_to_pos(v)
                assert np.all(c <= np.array(p, np.float32)) and np.all(np.array(p, np.float32) < c + size)
                assert np.allclose((c / size), np.round(c / size))          # corners sit on the voxel lattice
    # map spans [-4, 4): outside -> 0xffffffff
    assert m.pos_to_voxel(4.0, 0, 0) == 0xffffffff and m.pos_to_voxel(0, -4.6, 0) == 0xffffffff
    assert m.pos_to_voxel(3.99, 3.99, 3.99) == (15 << 8 | 15 << 4 | 15)
    # (p - min) * recip in (-1, 0) truncates to index 0 and is accepted (float -> uint32 cast, operations.h:867)
    assert m.pos_to_voxel(-4.2, -4.0, -4.0) == 0
    assert m.pos_to_voxel(-4.5, -4.0, -4.0) == 0xffffffff                   # exactly -1.0 and below: rejected
    assert m.pos_to_voxel(float("nan"), 0, 0) == 0xffffffff


def test_pdf_lut():
    """(3) queryNormalPDF (basic_algorithms.h:417-422, 456-460): 1/sqrt(pi) normaliser, no 1/sigma, clamp."""
    m = make(kc.K0, kc.PARAMS)
    assert abs(m.query_pdf(0.0, 0.0, 1.0) - 0.56418955) < 1e-7
    assert abs(m.query_pdf(0.3, 0.3, 0.05) - 0.56418955) < 1e-7            # independent of sigma at the mean
    assert m.query_pdf(10.0, 0.0, 1.0) == F(1e-9) and m.query_pdf(-9.95, 0.0, 1.0) == F(1e-9)
    assert m.query_pdf(9.9, 0.0, 1.0) != F(1e-9)                            # |x| == 9.9 still indexes the table
    t = m.pdf_table()
    assert t.shape == (20000,) and abs(float(t[10000 + 1000]) - 0.56418955 * np.exp(-0.5)) < 1e-6
    assert m.query_pdf(1.0, 0.0, 1.0) == t[11000] and m.query_pdf(0.0, 0.5, 0.25) == t[8000]
    assert m.query_pdf(0.0, 0.0, 0.0) == F(1e-9)                            # PINNED: 0/0 -> 1e-9f


def test_forgetting_table():
    """(4) getForgettingFactor (basic_algorithms.h:32-48): 2.5^(-i/rate), 0 from max_forget_count on."""
    p = dict(kc.PARAMS, forgetting_rate=2.0, max_forget_count=3)
    m = make(kc.K0, p)
    for i in range(3):
        assert m.forgetting_factor(i) == F(2.5 ** (-i / 2.0))
    assert m.forgetting_factor(3) == 0.0 and m.forgetting_factor(4) == 0.0 and m.forgetting_factor(5) == 0.0
    m2 = make(kc.K0, dict(kc.PARAMS, forgetting_rate=1.0, max_forget_count=5))
    assert m2.forgetting_factor(4) == F(2.5 ** -4.0) and m2.forgetting_factor(5) == 0.0


def test_insert_helpers():
    """(8) addParticleByGlobalPos through the oracle's helper: slot order, full, outside."""
    m = make(kc.K0, kc.PARAMS)
    m.set_global_time_stamp(1)
    v = kc.voxel_of(kc.K0, 0.6, 0.1, 2.1)
    got = [m.add_particle(0.6, 0.1, 2.1, label=k) for k in range(9)]
    assert got[:7] == [v * 8 + s for s in range(1, 8)] and got[7] == 0xffffffff and got[8] == 0xffffffff
    assert m.add_particle(40.0, 0, 0) == 0xffffffff


def test_noise_cursor_preincrement():
    """queryNormalRandomPresetSD pre-increments: the first value used is table[1] (basic_algorithms.h:426-433)."""
    noise = np.arange(1000000, dtype=np.float32) * 1e-6
    p = dict(kc.PARAMS, nb_ptc_num_per_point=2)
    m = orc.OracleMap(dict(kc.K0, bin_order=1), p, noise)
    m.load_state(kc.empty_state(kc.K0))
    m.set_ring_state(kc.ring0())
    depth, cloud = kc.blank_frame(kc.K0)
    kc.set_point(cloud, kc.K0, 0, 0, (0.6, 0.1, 2.1), 0.5)
    q_back = np.array([0, 0, 1, 0], np.float32)
    m.update(depth, cloud, kc.ORIGIN, q_back, stop_after="birth")
    st = m.dump_state()
    idx = np.flatnonzero(st["status"] == 2)
    assert len(idx) == 2
    xs = sorted(st["px"][idx])
    assert xs[0] == F(F(0.6) + F(0.5) * noise[1]) and xs[1] == F(F(0.6) + F(0.5) * noise[4])
    assert m.ring_state()["birth_cursor"] == 6
