"""Independent replays, written from the REFERENCE TEXT (not from oracle/orb_cpu.cpp), of the two pieces of ORBextractor with
the most intricate sequential semantics, run against the oracle on hundreds of random inputs:

  * DistributeOctTree + ExtractorNode::DivideNode (reference src/ORBextractor.cc:481-763): a std::list replayed as a
    Python list with push_front / erase, the full pass, the `size + 3 nToExpand > N` switch to the expand-largest-first
    phase, its sort over (size, node) pairs, the early break at N nodes and the best-response pick;
  * the cell loop of ComputeKeyPointsOctTree (:765-823): cell geometry, clamping, the `continue` tests, the minThFAST retry
    of an empty cell, the offsets added to the cell's corners.

The one place the reference is not reproducible is the tie-break of `sort(vector<pair<int, ExtractorNode*>>)` on equal
sizes: heap addresses.  The replay takes the node's creation sequence number as its address (later node = higher address),
which is the rule DESIGN.md documents for the oracle and the GPU path.

The oracle is parity-unpinned (the reference holds no fixtures and cannot be built here); these replays are what stands in
for fixtures on this part of the path."""
import math

import numpy as np
import pytest

from eao_fusion_amd import synth


class _Node:
    __slots__ = ("UL", "UR", "BL", "BR", "keys", "no_more", "seq")
    _count = 0

    def __init__(self):
        self.keys = []
        self.no_more = False
        _Node._count += 1
        self.seq = _Node._count       # stands in for the heap address of the list node (see the module docstring)


def _f32(x):
    return np.float32(x)


def _divide(n):
    """ExtractorNode::DivideNode (:481-537)."""
    half_x = int(math.ceil(_f32(n.UR[0] - n.UL[0]) / _f32(2)))
    half_y = int(math.ceil(_f32(n.BR[1] - n.UL[1]) / _f32(2)))
    n1, n2, n3, n4 = _Node(), _Node(), _Node(), _Node()
    n1.UL = n.UL; n1.UR = (n.UL[0] + half_x, n.UL[1]); n1.BL = (n.UL[0], n.UL[1] + half_y); n1.BR = (n.UL[0] + half_x, n.UL[1] + half_y)
    n2.UL = n1.UR; n2.UR = n.UR; n2.BL = n1.BR; n2.BR = (n.UR[0], n.UL[1] + half_y)
    n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = n.BL; n3.BR = (n1.BR[0], n.BL[1])
    n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = n.BR
    for kp in n.keys:
        x, y = kp[1], kp[2]
        if x < n1.UR[0]:
            (n1 if y < n1.BR[1] else n3).keys.append(kp)
        elif y < n1.BR[1]:
            n2.keys.append(kp)
        else:
            n4.keys.append(kp)
    for c in (n1, n2, n3, n4):
        if len(c.keys) == 1:
            c.no_more = True
    return n1, n2, n3, n4


def distribute_replay(xyr, min_x, max_x, min_y, max_y, N):
    """DistributeOctTree (:539-763).  xyr: (n, 3) float32 (x, y, response).  Returns the indices of the selected keypoints
    in output order."""
    n_ini = int(round(float(_f32(max_x - min_x) / _f32(max_y - min_y))))     # C round(): halves away from zero; ratios here are not x.5
    hx = _f32(max_x - min_x) / _f32(n_ini)
    nodes = []                                # the std::list, front = index 0
    ini = []
    for i in range(n_ini):
        ni = _Node()
        ni.UL = (int(hx * _f32(i)), 0); ni.UR = (int(hx * _f32(i + 1)), 0)
        ni.BL = (ni.UL[0], max_y - min_y); ni.BR = (ni.UR[0], max_y - min_y)
        nodes.append(ni)
        ini.append(ni)
    for k in range(len(xyr)):
        kp = (k, _f32(xyr[k, 0]), _f32(xyr[k, 1]), _f32(xyr[k, 2]))
        ini[int(kp[1] / hx)].keys.append(kp)
    keep = []
    for nd in nodes:
        if len(nd.keys) == 1:
            nd.no_more = True
            keep.append(nd)
        elif len(nd.keys) > 0:
            keep.append(nd)
    nodes = keep
    finish = False
    size_and_node = []
    while not finish:
        prev_size = len(nodes)
        n_to_expand = 0
        size_and_node = []
        i = 0
        while i < len(nodes):
            nd = nodes[i]
            if nd.no_more:
                i += 1
                continue
            for c in _divide(nd):
                if len(c.keys) > 0:
                    nodes.insert(0, c)            # push_front
                    i += 1                        # the iterator still points at `nd`
                    if len(c.keys) > 1:
                        n_to_expand += 1
                        size_and_node.append((len(c.keys), c))
            del nodes[i]                          # lit = lNodes.erase(lit)
        if len(nodes) >= N or len(nodes) == prev_size:
            finish = True
        elif len(nodes) + n_to_expand * 3 > N:
            while not finish:
                prev_size = len(nodes)
                prev = sorted(size_and_node, key=lambda sn: (sn[0], sn[1].seq))
                size_and_node = []
                for j in range(len(prev) - 1, -1, -1):
                    nd = prev[j][1]
                    for c in _divide(nd):
                        if len(c.keys) > 0:
                            nodes.insert(0, c)
                            if len(c.keys) > 1:
                                size_and_node.append((len(c.keys), c))
                    nodes.remove(nd)              # lNodes.erase(node->lit) (identity comparison: no __eq__ on _Node)
                    if len(nodes) >= N:
                        break
                if len(nodes) >= N or len(nodes) == prev_size:
                    finish = True
    out = []
    for nd in nodes:
        best = nd.keys[0]
        for kp in nd.keys[1:]:
            if kp[3] > best[3]:
                best = kp
        out.append(best[0])
    return out


def _random_candidates(rng, w, h, n, clustered):
    if clustered:      # a few dense blobs + background: forces deep, uneven trees and many equal-size ties
        k = rng.integers(2, 7)
        cx, cy = rng.uniform(0, w, k), rng.uniform(0, h, k)
        which = rng.integers(0, k, n)
        x = np.clip(cx[which] + rng.normal(0, w / 25, n), 0, w - 1)
        y = np.clip(cy[which] + rng.normal(0, h / 25, n), 0, h - 1)
    else:
        x, y = rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)
    # FAST corners have integer coordinates and integer responses (many response ties)
    xy = np.unique(np.stack([np.floor(x), np.floor(y)], 1), axis=0)
    rng.shuffle(xy)
    # upstream's order: cells row-major, corners row-major inside a cell -- any order is legal input for the routine itself;
    # sort row-major (the common case) for half of the sets
    if rng.random() < 0.5:
        xy = xy[np.lexsort((xy[:, 0], xy[:, 1]))]
    r = rng.integers(7, 120, len(xy)).astype(np.float32)
    return np.concatenate([xy.astype(np.float32), r[:, None]], 1)


@pytest.mark.parametrize("seed", range(12))
def test_distribute_octree_against_independent_replay(oracle, seed):
    """20 random candidate sets per seed (240 in all): box aspect ratios that give 1, 2, 3 and 4 initial nodes (640x480,
    752x480, the 1241x376 of KITTI -- three initial nodes), from a handful to thousands of candidates, quotas from 1 up."""
    rng = np.random.default_rng(9000 + seed)
    boxes = [(608, 448), (501, 368), (147, 102), (720, 448), (1209, 344), (1209, 300), (300, 300), (90, 61)]
    for _ in range(20):
        bw, bh = boxes[rng.integers(0, len(boxes))]
        n = int(rng.choice([1, 2, 5, 30, 200, 900, 3000]))
        N = int(rng.choice([1, 3, 17, 60, 217, 500, 1000]))
        xyr = _random_candidates(rng, bw, bh, n, clustered=rng.random() < 0.5)
        got = list(oracle.distribute(xyr, 16, 16 + bw, 16, 16 + bh, N))
        want = distribute_replay(xyr, 16, 16 + bw, 16, 16 + bh, N)
        assert got == want, "box %dx%d, %d candidates, N %d" % (bw, bh, len(xyr), N)


def cells_replay(oracle, img, ini_th, min_th):
    """The cell loop of ComputeKeyPointsOctTree (:773-823) over one level image (without the 19-px border: the level Mat of
    the reference is the ROI of its bordered buffer).  cv::FAST on the sub-image is the oracle's restatement of OpenCV's
    routine (pinned separately by a numpy replay in test_oracle_orb.py)."""
    EDGE_THRESHOLD, W = 19, np.float32(30)
    rows, cols = img.shape
    min_bx = EDGE_THRESHOLD - 3; min_by = min_bx
    max_bx = cols - EDGE_THRESHOLD + 3; max_by = rows - EDGE_THRESHOLD + 3
    width = np.float32(max_bx - min_bx); height = np.float32(max_by - min_by)
    n_cols = int(width / W); n_rows = int(height / W)
    w_cell = int(math.ceil(width / np.float32(n_cols))); h_cell = int(math.ceil(height / np.float32(n_rows)))
    out = []
    for i in range(n_rows):
        ini_y = np.float32(min_by + i * h_cell)
        max_y = ini_y + np.float32(h_cell + 6)
        if ini_y >= max_by - 3:
            continue
        if max_y > max_by:
            max_y = np.float32(max_by)
        for j in range(n_cols):
            ini_x = np.float32(min_bx + j * w_cell)
            max_x = ini_x + np.float32(w_cell + 6)
            if ini_x >= max_bx - 6:
                continue
            if max_x > max_bx:
                max_x = np.float32(max_bx)
            sub = img[int(ini_y):int(max_y), int(ini_x):int(max_x)]
            kps = oracle.fast(sub, ini_th)
            if len(kps) == 0:
                kps = oracle.fast(sub, min_th)
            for x, y, r in kps:
                out.append((np.float32(x) + np.float32(j * w_cell), np.float32(y) + np.float32(i * h_cell), r))
    return np.array(out, np.float32).reshape(-1, 3), (min_bx, max_bx, min_by, max_by)


@pytest.mark.parametrize("case", [dict(seed=1000), dict(seed=7, n_rect=40, n_small=0), dict(seed=1003, w=752, h=480),
                                  dict(seed=1004, w=1241, h=376), dict(seed=1005, w=322, h=241)])
def test_cell_loop_and_distribution_against_independent_replay(oracle, case):
    """Level by level: the replayed cell loop gives the oracle's candidate list (order included), and the replayed
    DistributeOctTree on that list gives the oracle's keypoints of the level (positions, responses, order)."""
    kw = dict(case)
    seed = kw.pop("seed")
    img = synth.synth_frame(seed, **kw)
    e = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    e.extract(img)
    quota = e.tables()["quota"]
    for l in range(8):
        lv = e.level_image(l)
        cand, (min_bx, max_bx, min_by, max_by) = cells_replay(oracle, lv, 20, 7)
        assert np.array_equal(cand, e.level_candidates(l)), "candidates of level %d" % l
        sel = distribute_replay(cand, min_bx, max_bx, min_by, max_by, int(quota[l]))
        lk = e.level_keypoints(l)
        assert len(sel) == len(lk)
        assert np.array_equal(cand[sel, 0] + np.float32(min_bx), lk["x"]) and np.array_equal(cand[sel, 1] + np.float32(min_by), lk["y"])
        assert np.array_equal(cand[sel, 2], lk["response"])
