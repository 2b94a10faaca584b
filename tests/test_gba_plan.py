"""The elimination order and launch schedule of the map-scale BundleAdjustment path (round 6: csrc/gba.hip gba_build_plan -- one level of nested dissection of the
keyframe graph, tile-level symbolic fill-in, list-scheduled panels, level-scheduled back substitution; the reference's counterpart is SimplicialLDLT behind an AMD
ordering, Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:95-112) checked WITHOUT a GPU: eao_bundle_adjustment_plan exports the plan, and a numpy replay executes it
launch by launch with the kernels' semantics -- every record reads its panel rows, solves them against the panel's factored diagonal block, archives, subtracts from its
target tile, look-ahead records factor the next diagonal block of their quadrant only -- on a random symmetric positive definite system with the pattern's blocks, then
the back substitution descriptor by descriptor.  The replay asserts the schedule's contracts (a diagonal block is factored before its panel runs, no two records of a
launch write one tile, nobody reads a tile the same launch writes, only live tiles are touched) and the solution must equal a dense solve."""
import ctypes as C

import numpy as np
import pytest

import eao_fusion_amd._lib as L


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_free", "n_rows", "n_tile_rows", "n_panels", "n_tiles", "n_segments", "n_separator", "separator_start", "rcm", "bandwidth",
                                         "chain_natural", "chain_estimate", "n_launches", "n_work", "n_diag", "n_sb", "n_sb_launches")]


def get_plan(n_free, pairs, force=0):
    lib = L.load()
    a = np.ascontiguousarray([p[0] for p in pairs], np.int32)
    b = np.ascontiguousarray([p[1] for p in pairs], np.int32)
    info = Info()
    none = C.c_void_p(None)
    st = lib.eao_bundle_adjustment_plan(n_free, len(pairs), a.ctypes.data, b.ctypes.data, force, C.byref(info), none, none, 0, none, 0, none, 0, none, 0, none, 0, none, 0)
    assert st == 0, lib.eao_last_error()
    T = info.n_tile_rows
    row_of = np.zeros(n_free, np.int32); tile_map = np.zeros(T * T, np.int32); work = np.zeros((max(info.n_work, 1), 8), np.int32)
    launches = np.zeros((max(info.n_launches, 1), 4), np.int32); diag = np.zeros(max(info.n_diag, 1), np.int32)
    sb = np.zeros((max(info.n_sb, 1), 4), np.int32); sbl = np.zeros((max(info.n_sb_launches, 1), 3), np.int32)
    st = lib.eao_bundle_adjustment_plan(n_free, len(pairs), a.ctypes.data, b.ctypes.data, force, C.byref(info), row_of.ctypes.data, tile_map.ctypes.data, tile_map.size,
                                        work.ctypes.data, len(work), launches.ctypes.data, len(launches), diag.ctypes.data, len(diag), sb.ctypes.data, len(sb),
                                        sbl.ctypes.data, len(sbl))
    assert st == 0, lib.eao_last_error()
    return dict(info=info, row_of=row_of, tile_map=tile_map.reshape(T, T), work=work[:info.n_work], launches=launches[:info.n_launches], diag=diag[:info.n_diag],
                sb=sb[:info.n_sb], sb_launches=sbl[:info.n_sb_launches])


def ldl(B):
    """unpivoted LDL^T of a symmetric block: unit lower L, d"""
    n = len(B)
    Lm, d = np.eye(n), np.zeros(n)
    A = B.copy()
    for p in range(n):
        d[p] = A[p, p]
        assert abs(d[p]) > 1e-12
        Lm[p + 1:, p] = A[p + 1:, p] / d[p]
        A[p + 1:, p + 1:] -= np.outer(Lm[p + 1:, p], A[p + 1:, p])
    return Lm, d


def replay(pl, pairs, seed):
    info, row_of, tm = pl["info"], pl["row_of"], pl["tile_map"]
    n, N, T = info.n_free, info.n_rows, info.n_tile_rows
    tN = N // 64
    assert N % 64 == 0 and T == tN + 1 and info.n_panels == N // 32
    # rows: every camera's six rows are distinct, inside [0, N)
    rows = np.concatenate([row_of[i] + np.arange(6) for i in range(n)])
    assert len(set(rows.tolist())) == 6 * n and rows.min() >= 0 and rows.max() < N
    rng = np.random.default_rng(seed)
    S = np.zeros((N, N))
    for (i, j) in pairs:
        blk = rng.normal(0, 1, (6, 6))
        ri, rj = row_of[i], row_of[j]
        if i == j:
            S[ri:ri + 6, ri:ri + 6] += blk @ blk.T
        else:
            S[ri:ri + 6, rj:rj + 6] += 0.3 * blk
            S[rj:rj + 6, ri:ri + 6] += 0.3 * blk.T
    real = np.zeros(N, bool); real[rows] = True
    S[np.arange(N), np.arange(N)] += np.where(real, np.abs(S).sum(1) + 1.0, 1.0)      # diagonally dominant; identity on the padding rows
    rhs = np.where(real, rng.normal(0, 1, N), 0.0)
    want = np.linalg.solve(S, rhs)
    # every non-zero of the lower triangle lies in a live tile
    nz = np.argwhere(np.tril(S) != 0)
    assert (tm[nz[:, 0] // 64, nz[:, 1] // 64] >= 0).all()
    A = np.zeros((T * 64, T * 64)); A[:N, :N] = np.tril(S); A[N, :N] = rhs
    Lar = np.zeros_like(A)
    dg = {}

    def blockfull(r0):      # the symmetric 32 x 32 block whose lower triangle sits at (r0, r0)
        Bq = A[r0:r0 + 32, r0:r0 + 32]
        return np.tril(Bq) + np.tril(Bq, -1).T

    for (off, cnt, doff, dcnt) in pl["launches"]:
        for kb in pl["diag"][doff:doff + dcnt]:
            assert kb not in dg
            dg[int(kb)] = ldl(blockfull(32 * kb))
        recs = pl["work"][off:off + cnt]
        targets = [int(r[2]) for r in recs if not (r[6] & 4)]
        assert len(targets) == len(set(targets)), "two records of one launch write the same tile"
        # nobody reads what the same launch changes: a record reads columns [k0, k0 + 32) of its two tile rows; a target (ta, tb) changes the columns of tile tb
        # whose ROWS lie below the writing panel (the others get a zero subtracted: the same bits go back)
        reads = set()
        for r in recs:
            for t_ in (int(r[0]), int(r[1])):
                reads.add((t_, 32 * int(r[5])))
        for r in recs:
            if r[6] & 4:
                continue
            c_lo = max(64 * int(r[1]), 32 * int(r[5]) + 32)
            for (t_, k0_) in reads:
                assert not (t_ == int(r[0]) and k0_ >= c_lo and k0_ < 64 * int(r[1]) + 64), "a launch reads columns it changes"
        arch = set()
        for (ti, tj, sc, sw, sl, kb, flags, nxt) in recs:
            kc, k0 = kb // 2, 32 * kb
            assert tm[ti, tj] == sc and sc >= 0 and tm[ti, kc] == sw and sw >= 0 and tm[tj, kc] == sl and sl >= 0 and ti >= tj
            assert kb in dg, "panel %d runs before its diagonal block is factored" % kb
            Lkk, d = dg[int(kb)]
            Li = np.linalg.inv(Lkk)

            def solve(t):
                r = 64 * t + np.arange(64)
                act = (r >= k0 + 32) & (r <= N)
                Wm = np.zeros((64, 32))
                Wm[act] = A[r[act], k0:k0 + 32] @ Li.T
                return r, act, Wm
            ri, acti, Wi = solve(ti)
            rj, actj, Wj = solve(tj)
            if flags & 2:
                assert (int(kb), int(ti)) not in arch
                arch.add((int(kb), int(ti)))
                Lar[ri[acti], k0:k0 + 32] = Wi[acti] / d
            if flags & 4:
                assert ti == tN and tj == tN
                continue
            upd = Wi @ (Wj / d).T
            if not (flags & 1):
                A[64 * ti:64 * ti + 64, 64 * tj:64 * tj + 64] -= upd
            else:
                assert ti == tj and nxt // 2 == ti and nxt not in dg
                oq = 32 * nxt - 64 * tj
                quads = [(oq, oq)] + ([(32, 0), (32, 32)] if oq == 0 else [])
                for (qr, qc) in quads:
                    A[64 * ti + qr:64 * ti + qr + 32, 64 * tj + qc:64 * tj + qc + 32] -= upd[qr:qr + 32, qc:qc + 32]
                dg[int(nxt)] = ldl(blockfull(32 * nxt))
        # every tile row a panel of this launch reaches has its l entries archived once
    assert sorted(dg) == list(range(N // 32)), "a panel's diagonal block was never factored"
    z = Lar[N, :N].copy()
    x = np.full(N, np.nan)
    Lfull = Lar[:N, :N].copy()
    for kb, (Lkk, _d) in dg.items():
        Lfull[32 * kb:32 * kb + 32, 32 * kb:32 * kb + 32] = Lkk
    for (off, cnt, _gx) in pl["sb_launches"]:
        written = []
        for (J0, w, lo, hi) in pl["sb"][off:off + cnt]:
            assert w % 32 == 0 and 0 < w <= 256 and J0 % 64 == 0 and J0 + w <= N and np.isnan(x[J0:J0 + w]).all()
            xl = np.linalg.solve(Lfull[J0:J0 + w, J0:J0 + w].T, z[J0:J0 + w])
            x[J0:J0 + w] = xl
            for c in range(lo, hi):
                written.append(c)
                cols = slice(64 * c, 64 * c + 64)
                assert 64 * c + 64 <= J0
                z[cols] -= Lfull[J0:J0 + w, cols].T @ xl
        assert len(written) == len(set(written)), "two super-blocks of one launch update the same columns"
    assert not np.isnan(x).any(), "a column was never solved"
    err = np.abs(x - want).max() / max(np.abs(want).max(), 1e-30)
    assert err < 1e-9, err
    return err


def band_pairs(n, band, ring=False):
    out = set()
    for i in range(n):
        out.add((i, i))
        for k in range(1, band + 1):
            j = i + k
            if j < n:
                out.add((i, j))
            elif ring:
                out.add((j - n, i))
    return sorted(out)


CASES = [
    ("band", 120, band_pairs(120, 5), 0), ("band forced 4", 120, band_pairs(120, 5), 4), ("band natural", 120, band_pairs(120, 5), 1), ("band forced 9", 150, band_pairs(150, 3), 9),
    ("ring", 130, band_pairs(130, 4, ring=True), 0), ("ring forced 6", 130, band_pairs(130, 4, ring=True), 6), ("dense", 40, [(i, j) for i in range(40) for j in range(i, 40)], 0),
    ("dense forced 3", 40, [(i, j) for i in range(40) for j in range(i, 40)], 3), ("tiny", 3, [(0, 0), (1, 1), (2, 2), (0, 2)], 0),
]


@pytest.mark.parametrize("name,n,pairs,force", CASES, ids=[c[0] for c in CASES])
def test_schedule_replays_to_the_dense_solution(name, n, pairs, force):
    pl = get_plan(n, pairs, force)
    replay(pl, pairs, seed=len(pairs) + force)


def test_random_graphs_and_components():
    rng = np.random.default_rng(7)
    for trial in range(6):
        n = int(rng.integers(60, 140))
        pairs = set((i, i) for i in range(n))
        # two or three trajectories that never see each other, a few loop closures, shuffled keyframe ids in half of the trials
        perm = rng.permutation(n) if trial % 2 else np.arange(n)
        comps = np.array_split(np.arange(n), int(rng.integers(1, 4)))
        for comp in comps:
            for a in range(len(comp)):
                for k in range(1, int(rng.integers(2, 6))):
                    if a + k < len(comp):
                        i, j = int(perm[comp[a]]), int(perm[comp[a + k]])
                        pairs.add((min(i, j), max(i, j)))
            for _ in range(2):
                i, j = int(perm[rng.choice(comp)]), int(perm[rng.choice(comp)])
                pairs.add((min(i, j), max(i, j)))
        pairs = sorted(pairs)
        for force in (0, 5):
            replay(get_plan(n, pairs, force), pairs, seed=100 + trial)


def check_structure(pl):
    """the schedule's contracts alone (no matrix): usable on maps too large for a dense replay"""
    info, tm = pl["info"], pl["tile_map"]
    N, tN = info.n_rows, info.n_rows // 64
    have = set()
    for (off, cnt, doff, dcnt) in pl["launches"]:
        for kb in pl["diag"][doff:doff + dcnt]:
            assert int(kb) not in have
            have.add(int(kb))
        recs = pl["work"][off:off + cnt]
        targets = [int(r[2]) for r in recs if not (r[6] & 4)]
        assert len(targets) == len(set(targets))
        reads = set((int(t_), 32 * int(r[5])) for r in recs for t_ in (r[0], r[1]))
        rows_read = {}
        for (t_, k0_) in reads:
            rows_read.setdefault(t_, []).append(k0_)
        new = set()
        for (ti, tj, sc, sw, sl, kb, flags, nxt) in recs:
            kc = kb // 2
            assert int(kb) in have and tm[ti, tj] == sc and sc >= 0 and tm[ti, kc] == sw and sw >= 0 and tm[tj, kc] == sl and sl >= 0
            if not (flags & 4):
                c_lo = max(64 * int(tj), 32 * int(kb) + 32)
                assert not any(c_lo <= k0_ < 64 * int(tj) + 64 for k0_ in rows_read.get(int(ti), []))
            if flags & 1:
                assert ti == tj and nxt // 2 == ti and int(nxt) not in have and int(nxt) not in new
                new.add(int(nxt))
        have |= new
    assert have == set(range(N // 32))
    cols = np.zeros(N, int)
    for (off, cnt, _gx) in pl["sb_launches"]:
        for (J0, w, lo, hi) in pl["sb"][off:off + cnt]:
            assert (cols[J0:J0 + w] == 0).all() and 64 * hi <= J0 and lo <= hi
            cols[J0:J0 + w] = 1
    assert cols.all()


def test_large_maps_keep_the_contracts():
    """2000 keyframes on a band with 40 loop closures, and 1500 keyframes of a random sparse graph in shuffled order: the schedule's contracts hold (no dense replay at this size)"""
    rng = np.random.default_rng(11)
    pairs = set(band_pairs(2000, 8))
    for _ in range(40):
        i, j = sorted(int(x) for x in rng.integers(0, 2000, 2))
        pairs.add((i, j))
    pl = get_plan(2000, sorted(pairs))
    check_structure(pl)
    assert pl["info"].n_launches < pl["info"].chain_natural // 3
    perm = rng.permutation(1500)
    pairs = set((i, i) for i in range(1500))
    for a in range(1500):
        for k in range(1, 6):
            if a + k < 1500:
                i, j = int(perm[a]), int(perm[a + k])
                pairs.add((min(i, j), max(i, j)))
    pl = get_plan(1500, sorted(pairs))
    check_structure(pl)
    assert pl["info"].rcm == 1 and pl["info"].n_launches < pl["info"].chain_natural      # (shuffled ids: the natural line is useless, reverse Cuthill-McKee finds the trajectory)


def test_a_trajectory_gets_a_short_chain():
    """1000 keyframes, each covisible with its +-10 neighbours (bench.py's banded map): the factorisation's chain of dependent launches falls from 188 panels to a few dozen"""
    pl = get_plan(1000, band_pairs(1000, 10))
    i = pl["info"]
    assert i.chain_natural >= 188 and i.n_segments >= 6 and i.n_launches <= 60 and i.n_sb_launches <= 12, (i.n_segments, i.n_launches, i.n_sb_launches, i.n_separator)
    # ... and a map in which every keyframe sees every other one stays in natural order
    d = get_plan(64, [(a, b) for a in range(64) for b in range(a, 64)])["info"]
    assert d.n_segments == 1 and d.n_separator == 0
