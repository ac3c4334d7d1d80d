"""The known-answer cases of tests/kat_cases.py run on the MI355X through the C ABI."""
import numpy as np
import pytest

from semantic_dsp_map_amd import binding
from tests import kat_cases as kc

pytestmark = pytest.mark.gpu


def make(cfg, params):
    return binding.SdmMap(cfg, params, np.zeros(1000000, np.float32))


@pytest.mark.parametrize("case", kc.ALL_CASES, ids=lambda c: c.__name__)
def test_case(case):
    case(make)


def test_weight_closed_form():
    m = make(kc.K0, kc.PARAMS)
    kc.case_weight_closed_form(make, m.download_pdf_table())


def test_create_rejects_bad_config():
    bad = dict(kc.K0, x_n=12, y_n=12, z_n=12)      # 36 + 3 bits > 31 (operations.h:54-58)
    with pytest.raises(binding.SdmError):
        binding.SdmMap(bad, kc.PARAMS)
    with pytest.raises(binding.SdmError):
        binding.SdmMap(kc.K0, kc.PARAMS, device=99)


def test_clear_resets_map_but_keeps_ring_motion():
    """SemanticDSPMap::clear / RingBufferOperations::clear (semantic_dsp_map.h:74-81, operations.h:683-723)."""
    m = make(kc.K0, kc.PARAMS)
    depth, cloud = kc.blank_frame(kc.K0)
    kc.set_point(cloud, kc.K0, 0, 0, (0.6, 0.1, 2.1), 0.2, track=5, label=3)
    m.update(depth, cloud, np.array([1.6, 0, 0], np.float32), kc.IDENT_Q, sync=True)
    assert (m.dump_state()["status"] == 2).sum() == 1
    m.clear()
    st = m.dump_state()
    assert np.all(st["status"].reshape(-1, 8)[:, 0] == 5) and np.all(st["status"].reshape(-1, 8)[:, 1:] == 0)
    assert not st["w"].any() and np.all(st["owner"] == 0xFFFF)
    rs = m.ring_state()
    assert rs["global_time_stamp"] == 0 and rs["moved_steps"] == [3, 0, 0]     # movement retained
    assert not any(s.any() for s in m.stamps())
