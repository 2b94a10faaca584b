"""GPU: the remaining guided searches (eao_search_by_projection_sim3 / _kf, eao_search_by_bow, eao_search_for_triangulation,
eao_search_for_initialization, eao_fuse_search, eao_search_by_sim3) against the CPU oracle -- bit-exact match tables."""
import numpy as np
import pytest

from eao_fusion_amd import search, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["host arrays", "keyframe handles"])
def both(request, oracle):
    """Every case below runs twice: through the host-array entry points (eao_search_* / eao_fuse_search*) and through keyframe handles (eao_kf_*, round 5: frames
    resident in HBM, the vocabulary-node searches and Fuse selected on the device) -- the same match tables as the oracle either way."""
    import torch  # noqa: F401  (first, so that the library resolves the same HIP runtime)
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return (search.product() if request.param == "host arrays" else search.product_handles()), oracle.search_binding()


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


def _neighbours(scene, n_nb):
    """n_nb variants of K2 as neighbour keyframes: shifted keypoints, other occupancy, another fundamental matrix scale / epipole, one with an empty feature
    vector overlap -- different work per neighbour, so a mixed-up slice of the shared distance array cannot pass."""
    out = []
    for k in range(n_nb):
        k2 = dict(scene["K2"])
        k2["kp_x"] = (scene["K2"]["kp_x"] + np.float32(0.25 * k)).astype(np.float32)
        k2["occupied"] = ((scene["mp2"] >= 0) & (np.arange(len(scene["mp2"])) % (3 + k) == 0)).astype(np.uint8)
        fv = dict(scene["fv2"])
        if k == 2:
            fv = dict(fv); fv["node_id"] = (fv["node_id"] + 100000).astype(np.uint32)      # no common vocabulary node with K1
        if k % 2:
            perm = np.random.default_rng(900 + k).permutation(len(k2["kp_x"]))              # another keypoint order (and descriptor order)
            for key in ("kp_x", "kp_y", "kp_octave", "kp_angle", "u_right", "descriptors", "occupied"):
                k2[key] = np.ascontiguousarray(k2[key][perm])
            inv = np.argsort(perm)
            fv = dict(fv); fv["index"] = inv[fv["index"]].astype(np.uint32)
        F = (scene["F12"] * np.float32(1.0 + 0.01 * k)).astype(np.float32)
        out.append((k2, fv, F, float(scene["ex"]) + k, float(scene["ey"]) - k))
    return out


@pytest.mark.parametrize("only_stereo", [0, 1])
def test_search_for_triangulation_batch_equals_single_calls(both, scene, only_stereo):
    """eao_search_for_triangulation_batch (LocalMapping::CreateNewMapPoints: one search per neighbour keyframe, src/LocalMapping.cc:211-290): row k of the batch
    = the single call against neighbour k = the oracle's, entry for entry; 0, 1 and 7 neighbours."""
    g, o = both
    k1 = dict(scene["K1"])
    k1["occupied"] = ((scene["mp1"] >= 0) & (np.arange(len(scene["mp1"])) % 2 == 0)).astype(np.uint8)
    for n_nb in (7, 1, 0):
        nb = _neighbours(scene, n_nb)
        nm, m = g.search_for_triangulation_batch(k1, scene["fv1"], [x[0] for x in nb], [x[1] for x in nb], [x[2] for x in nb], [x[3] for x in nb],
                                                 [x[4] for x in nb], only_stereo, True)
        assert m.shape == (n_nb, len(k1["kp_x"]))
        for k, (k2, fv, F, ex, ey) in enumerate(nb):
            want = o.search_for_triangulation(k1, scene["fv1"], k2, fv, F, ex, ey, only_stereo, True)
            single = g.search_for_triangulation(k1, scene["fv1"], k2, fv, F, ex, ey, only_stereo, True)
            assert nm[k] == want[0] == single[0] and np.array_equal(m[k], want[1]) and np.array_equal(single[1], want[1]), "neighbour %d" % k
        if n_nb == 7:
            assert nm[2] == 0 and nm.sum() > 20


@pytest.mark.parametrize("use_sim3", [0, 1])
def test_fuse_search_batch_equals_single_calls(both, scene, use_sim3):
    """eao_fuse_search_batch (LocalMapping::SearchInNeighbors: the same points into every target keyframe, src/LocalMapping.cc:458-520) against single calls and the oracle."""
    g, o = both
    kfs, poses = [], []
    for k in range(5):
        kf = dict(scene["K2"])
        kf["kp_y"] = (scene["K2"]["kp_y"] + np.float32(0.5 * k)).astype(np.float32)
        kfs.append(kf)
        if use_sim3:
            S = scene["Scw"].copy(); S[0, 3] += np.float32(0.002 * k)
            poses.append(S.ravel())
        else:
            T = scene["T2w"].astype(np.float64).copy(); T[0, 3] += 0.002 * k
            poses.append(np.concatenate([T[:3, :3].ravel(), T[:3, 3], -T[:3, :3].T @ T[:3, 3]]).astype(np.float32))
    nf, best = g.fuse_search_batch(kfs, use_sim3, poses, scene["K"], scene["bf"], scene["points"], 3.0)
    for k in range(5):
        want = o.fuse_search(kfs[k], use_sim3, poses[k], scene["K"], scene["bf"], scene["points"], 3.0)
        assert nf[k] == want[0] and np.array_equal(best[k], want[1]), "target %d" % k
    assert nf.sum() > 30


def test_searches_refuse_bad_input(both, scene):
    """Negative cases of the host half: an octave beyond the frame's levels, NaN in a pose / fundamental matrix, a feature-vector index beyond
    the frame -- EAO_ERR_INVALID (an exception here), never a fault."""
    import eao_fusion_amd as E
    g, _ = both
    k2 = dict(scene["K2"]); k2["kp_octave"] = scene["K2"]["kp_octave"].copy(); k2["kp_octave"][5] = 8
    with pytest.raises(E.EaoError):
        g.search_for_triangulation(scene["K1"], scene["fv1"], k2, scene["fv2"], scene["F12"], scene["ex"], scene["ey"], 0, True)
    Fn = scene["F12"].copy(); Fn[1, 1] = np.nan
    with pytest.raises(E.EaoError):
        g.search_for_triangulation(scene["K1"], scene["fv1"], scene["K2"], scene["fv2"], Fn, scene["ex"], scene["ey"], 0, True)
    fv = dict(scene["fv2"]); fv["index"] = fv["index"].copy(); fv["index"][3] = len(scene["K2"]["kp_x"])
    with pytest.raises(E.EaoError):
        g.search_for_triangulation(scene["K1"], scene["fv1"], scene["K2"], fv, scene["F12"], scene["ex"], scene["ey"], 0, True)
    Sn = scene["Scw"].copy(); Sn[2, 3] = np.inf
    with pytest.raises(E.EaoError):
        g.fuse_search(scene["K2"], 1, Sn, scene["K"], scene["bf"], scene["points"], 3.0)
    with pytest.raises(E.EaoError):
        g.fuse_search(k2, 1, scene["Scw"], scene["K"], scene["bf"], scene["points"], 3.0)


def test_keyframe_handle_lifecycle(oracle):
    """Handles are reusable across searches and calls, their occupancy can be replaced (eao_keyframe_update_points), big vocabulary nodes (more than 64 features on
    either side: the in-register chunk and the memory rounds of the node kernel) and more neighbours than one launch carries (> 16) give the oracle's tables."""
    import torch  # noqa: F401
    g, o = search.product_handles(), oracle.search_binding()
    sc = synth.synth_search_scene(n=1500, seed=8110, n_nodes=9)          # ~190 features per node
    k1, k2 = dict(sc["K1"]), dict(sc["K2"])
    h1, h2 = g.handle(k1, sc["fv1"]), g.handle(k2, sc["fv2"])
    assert g.lib.eao_keyframe_size(h1.h) == len(k1["kp_x"])
    s1 = dict(descriptors=k1["descriptors"], angle=k1["kp_angle"], valid=(sc["mp1"] >= 0).astype(np.uint8), fv=sc["fv1"])
    s2 = dict(descriptors=k2["descriptors"], angle=k2["kp_angle"], valid=(sc["mp2"] >= 0).astype(np.uint8), fv=sc["fv2"])
    for mode in (0, 1):
        for rep in range(2):
            got = g.search_by_bow_h(mode, h1, s1["valid"], h2, s2["valid"], 0.8, True)
            want = o.search_by_bow(mode, s1, s2, 0.8, True)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and got[0] > 10
    for rnd in range(3):
        occ1 = (np.arange(len(k1["kp_x"])) % (2 + rnd) == 0).astype(np.uint8)
        occ2 = (np.arange(len(k2["kp_x"])) % (3 + rnd) == 1).astype(np.uint8)
        h1.update_points(occ1); h2.update_points(occ2 if rnd else None)
        a, b = dict(k1), dict(k2)
        a["occupied"], b["occupied"] = occ1, (occ2 if rnd else np.zeros_like(occ2))
        nm, m = g.search_for_triangulation_h(h1, [h2] * 19, [sc["F12"]] * 19, [sc["ex"]] * 19, [sc["ey"]] * 19, 0, True)
        want = o.search_for_triangulation(a, sc["fv1"], b, sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True)
        assert all(nm[k] == want[0] and np.array_equal(m[k], want[1]) for k in range(19)), "round %d" % rnd
    # a feature vector that files a keypoint under two nodes: the handle path hands over to the host replay (upstream's walk order matters across nodes)
    fv = {k: v.copy() for k, v in sc["fv2"].items()}
    fv["index"][fv["node_start"][1]] = fv["index"][fv["node_start"][0]]
    hd = g.handle(k2, fv)
    s2d = dict(s2); s2d["fv"] = fv
    got = g.search_by_bow_h(0, h1, s1["valid"], hd, None, 0.8, True)
    want = o.search_by_bow(0, s1, s2d, 0.8, True)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    T = sc["T2w"].astype(np.float64)
    pose = np.concatenate([T[:3, :3].ravel(), T[:3, 3], -T[:3, :3].T @ T[:3, 3]]).astype(np.float32)
    nf, best = g.fuse_search_h([h2] * 18, 0, [pose] * 18, sc["K"], sc["bf"], sc["points"], 3.0)
    want = o.fuse_search(k2, 0, pose, sc["K"], sc["bf"], sc["points"], 3.0)
    assert all(nf[k] == want[0] and np.array_equal(best[k], want[1]) for k in range(18)) and want[0] > 10
