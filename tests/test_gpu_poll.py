"""GPU: the two hand-overs that rest on "a PCIe read pushes posted writes" (VERDICT r4 weak #11, next #7b) re-checked on every driver box:
  * the tracker's done word (csrc/pose.hip pose_publish, csrc/track.hip): 20 000 eao_tracker_track_local_map calls in which consecutive calls on a handle differ
    -- tools/stress_track_poll.py, which compares every call bit for bit with the first result of its (scene, variant) and those with the oracle chain;
  * the keyframe-handle searches' done word (csrc/keyframe.hip k_kf_finish, round 5): 20 000 SearchByBoW / SearchForTriangulation calls alternating between
    problems of different size on one thread's context, every table compared with the first one its problem produced (those with the oracle).
A runtime / firmware change that breaks the assumption shows up here as a stale table, not in a user's map."""
import os
import subprocess
import sys

import numpy as np
import pytest

from eao_fusion_amd import search, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tracker_done_word_poll_stress():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_track_poll.py"), "20000", "4", "23"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, EAO_TRACK_POLL="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "20000 calls" in out.stdout and " 0 mismatches" in out.stdout, out.stdout[-800:]


def test_keyframe_done_word_poll_stress(oracle):
    import torch  # noqa: F401
    g, o = search.product_handles(), oracle.search_binding()
    rng = np.random.default_rng(29)
    probs = []
    for n, seed, nodes in ((300, 8501, 12), (1100, 8502, 60), (700, 8503, 35)):
        sc = synth.synth_search_scene(n=n, seed=seed, n_nodes=nodes)
        h1, h2 = g.handle(sc["K1"], sc["fv1"]), g.handle(sc["K2"], sc["fv2"])
        v1, v2 = (sc["mp1"] >= 0).astype(np.uint8), (sc["mp2"] >= 0).astype(np.uint8)
        s1 = dict(descriptors=sc["K1"]["descriptors"], angle=sc["K1"]["kp_angle"], valid=v1, fv=sc["fv1"])
        s2 = dict(descriptors=sc["K2"]["descriptors"], angle=sc["K2"]["kp_angle"], valid=v2, fv=sc["fv2"])
        calls = [lambda h1=h1, h2=h2, v1=v1: g.search_by_bow_h(0, h1, v1, h2, None, 0.75, True),
                 lambda h1=h1, h2=h2, v1=v1, v2=v2: g.search_by_bow_h(1, h1, v1, h2, v2, 0.8, True),
                 lambda h1=h1, h2=h2, sc=sc: g.search_for_triangulation_h(h1, [h2, h2, h2], [sc["F12"]] * 3, [sc["ex"]] * 3, [sc["ey"]] * 3, 0, True)]
        want = [o.search_by_bow(0, s1, s2, 0.75, True), o.search_by_bow(1, s1, s2, 0.8, True),
                o.search_for_triangulation(sc["K1"], sc["fv1"], sc["K2"], sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True)]
        for k, fn in enumerate(calls):
            first = fn()
            if k < 2:
                assert first[0] == want[k][0] and np.array_equal(first[1], want[k][1])
            else:
                assert all(first[0][q] == want[k][0] and np.array_equal(first[1][q], want[k][1]) for q in range(3))
            probs.append((fn, (np.array(first[0], copy=True), first[1].copy())))
    bad = 0
    for it in range(20000):
        fn, (n0, t0) = probs[int(rng.integers(len(probs)))]
        n, t = fn()
        if not (np.array_equal(np.asarray(n), n0) and np.array_equal(t, t0)):
            bad += 1
    assert bad == 0, "%d of 20000 polled calls returned a table that differs from their problem's first result" % bad
