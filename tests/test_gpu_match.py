"""GPU parity: guided matching (eao_search_by_projection_*) vs the sequential oracle -- identical assignments."""
import numpy as np
import pytest

from eao_fusion_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return E


@pytest.mark.parametrize("n,seed,th", [(1000, 7000, 1.0), (1000, 7003, 3.0), (200, 7004, 5.0), (3000, 7005, 1.0)])
def test_search_by_projection_points(gpu, oracle, n, seed, th):
    cur, last, mps = synth.synth_tracking(n=n, seed=seed)
    m = gpu.ORBmatcher(0.8, True)
    nm, got = m.SearchByProjectionPoints(cur, mps, th)
    onm, ref = oracle.search_by_projection_points(cur, mps, th, 0.8)
    assert nm == onm and np.array_equal(got, ref)
    assert nm > 0.3 * n


@pytest.mark.parametrize("n,seed,th,mono,moved", [(1000, 7000, 7.0, False, 0.03), (1000, 7006, 15.0, False, 0.3),
                                                  (1000, 7007, 7.0, False, -0.3), (500, 7008, 7.0, True, 0.03)])
def test_search_by_projection_frames(gpu, oracle, n, seed, th, mono, moved):
    cur, last, mps = synth.synth_tracking(n=n, seed=seed, moved=moved)   # |moved| > mb exercises the forward / backward level windows
    for check in (True, False):
        m = gpu.ORBmatcher(0.9, check)
        nm, got = m.SearchByProjectionFrames(cur, last, th, mono)
        onm, ref = oracle.search_by_projection_frames(cur, last, th, mono, check)
        assert nm == onm and np.array_equal(got, ref)
    assert nm > 0.2 * n


def test_search_edge_cases(gpu, oracle):
    cur, last, mps = synth.synth_tracking(n=100, seed=7009)
    m = gpu.ORBmatcher(0.8, True)
    mps["skip"][:] = 1                                   # nothing in view
    nm, got = m.SearchByProjectionPoints(cur, mps, 1.0)
    assert nm == 0 and (got == -1).all()
    last["valid"][:] = 0
    nm, got = m.SearchByProjectionFrames(cur, last, 7.0, False)
    assert nm == 0 and (got == -1).all()
    mps["skip"][:] = 0
    mps["proj_x"][:] = 5000.0                            # window entirely outside the grid: early return of GetFeaturesInArea
    nm, got = m.SearchByProjectionPoints(cur, mps, 1.0)
    assert nm == 0
