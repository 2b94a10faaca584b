"""GPU: the remaining guided searches (eao_search_by_projection_sim3 / _kf, eao_search_by_bow, eao_search_for_triangulation,
eao_search_for_initialization, eao_fuse_search, eao_search_by_sim3) against the CPU oracle -- bit-exact match tables."""
import numpy as np
import pytest

from eao_fusion_amd import search, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def both(oracle):
    import torch  # noqa: F401  (first, so that the library resolves the same HIP runtime)
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return search.product(), oracle.search_binding()


@pytest.fixture(scope="module", params=[dict(), dict(n=1200, seed=8001, flip=0.09, mono_frac=0.6), dict(n=300, seed=8002, clutter=0.5, n_nodes=12)])
def scene(request):
    return synth.synth_search_scene(**request.param)


def _same(a, b):
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)
    return a[0]


def _pts_of(scene, mp):
    P = scene["points"]
    idx = np.maximum(mp, 0)
    d = {k: np.ascontiguousarray(P[k][idx]) for k in ("Xw", "normal", "min_dist_inv", "max_dist_inv", "max_dist", "descriptors")}
    d["active"] = ((mp >= 0) & (P["active"][idx] > 0)).astype(np.uint8)
    return d


def test_search_by_projection_sim3(both, scene):
    g, o = both
    kf = dict(scene["K2"])
    kf["occupied"] = (np.arange(len(kf["kp_x"])) % 13 == 0).astype(np.uint8)
    for th in (10, 4):
        n = _same(g.search_by_projection_sim3(kf, scene["Scw"], scene["K"], scene["points"], th),
                  o.search_by_projection_sim3(kf, scene["Scw"], scene["K"], scene["points"], th))
    assert n > 10


def test_search_by_projection_kf(both, scene):
    g, o = both
    cur = dict(scene["K2"])
    cur["occupied"] = (np.arange(len(cur["kp_x"])) % 11 == 0).astype(np.uint8)
    P = scene["points"]
    ang = ((np.arange(len(P["active"])) * 37) % 360).astype(np.float32)
    for check, th, od in ((True, 15, 100), (False, 10, 64)):
        _same(g.search_by_projection_kf(cur, scene["T2w"], scene["K"], P, ang, th, od, check),
              o.search_by_projection_kf(cur, scene["T2w"], scene["K"], P, ang, th, od, check))


def _sides(scene):
    def side(K, mp, fv):
        return dict(descriptors=K["descriptors"], angle=K["kp_angle"], valid=(mp >= 0).astype(np.uint8), fv=fv)
    return side(scene["K1"], scene["mp1"], scene["fv1"]), side(scene["K2"], scene["mp2"], scene["fv2"])


@pytest.mark.parametrize("mode", [0, 1])
def test_search_by_bow(both, scene, mode):
    g, o = both
    s1, s2 = _sides(scene)
    for ratio, check in ((0.7, True), (0.9, False)):
        n = _same(g.search_by_bow(mode, s1, s2, ratio, check), o.search_by_bow(mode, s1, s2, ratio, check))
    assert n > 10


def test_search_by_bow_disjoint_vocabulary(both, scene):
    g, o = both
    s1, s2 = _sides(scene)
    s2 = dict(s2)
    fv = dict(s2["fv"])
    fv["node_id"] = (fv["node_id"] + 100000).astype(np.uint32)     # no common node
    s2["fv"] = fv
    assert _same(g.search_by_bow(0, s1, s2, 0.7, True), o.search_by_bow(0, s1, s2, 0.7, True)) == 0


@pytest.mark.parametrize("only_stereo", [0, 1])
def test_search_for_triangulation(both, scene, only_stereo):
    g, o = both
    k1, k2 = dict(scene["K1"]), dict(scene["K2"])
    k1["occupied"] = ((scene["mp1"] >= 0) & (np.arange(len(scene["mp1"])) % 2 == 0)).astype(np.uint8)
    k2["occupied"] = ((scene["mp2"] >= 0) & (np.arange(len(scene["mp2"])) % 3 == 0)).astype(np.uint8)
    args = (k1, scene["fv1"], k2, scene["fv2"], scene["F12"], scene["ex"], scene["ey"], only_stereo, True)
    _same(g.search_for_triangulation(*args), o.search_for_triangulation(*args))


def test_search_for_initialization(both, scene):
    g, o = both
    f1, f2 = scene["K1"], scene["K2"]
    pm = np.stack([f1["kp_x"], f1["kp_y"]], 1)
    for window, ratio, check in ((100, 0.9, True), (40, 0.8, False)):
        _same(g.search_for_initialization(f1, f2, pm, window, ratio, check), o.search_for_initialization(f1, f2, pm, window, ratio, check))


@pytest.mark.parametrize("use_sim3", [0, 1])
def test_fuse_search(both, scene, use_sim3):
    g, o = both
    if use_sim3:
        pose = scene["Scw"]
    else:
        T = scene["T2w"].astype(np.float64)
        pose = np.concatenate([T[:3, :3].ravel(), T[:3, 3], -T[:3, :3].T @ T[:3, 3]]).astype(np.float32)
    n = _same(g.fuse_search(scene["K2"], use_sim3, pose, scene["K"], scene["bf"], scene["points"], 3.0),
              o.fuse_search(scene["K2"], use_sim3, pose, scene["K"], scene["bf"], scene["points"], 3.0))
    assert n > 10


def test_search_by_sim3(both, scene):
    g, o = both
    P1, P2 = _pts_of(scene, scene["mp1"]), _pts_of(scene, scene["mp2"])
    args = (scene["K1"], scene["T1w"], P1, scene["K2"], scene["T2w"], P2, scene["K"], 1.0, scene["R12"], scene["t12"], 7.5)
    assert _same(g.search_by_sim3(*args), o.search_by_sim3(*args)) > 10


def test_nothing_active(both, scene):
    g, o = both
    P = dict(scene["points"])
    P["active"] = np.zeros_like(P["active"])
    assert _same(g.search_by_projection_sim3(scene["K2"], scene["Scw"], scene["K"], P, 10),
                 o.search_by_projection_sim3(scene["K2"], scene["Scw"], scene["K"], P, 10)) == 0
    assert _same(g.fuse_search(scene["K2"], 1, scene["Scw"], scene["K"], scene["bf"], P, 3.0),
                 o.fuse_search(scene["K2"], 1, scene["Scw"], scene["K"], scene["bf"], P, 3.0)) == 0
