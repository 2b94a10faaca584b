"""GPU parity: Hamming kernels vs the oracle (bit-exact), BASELINE configs[2] = 1000 x 1000 descriptors."""
import numpy as np
import pytest

from eao_fusion_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return E


@pytest.mark.parametrize("na,nb", [(1000, 1000), (1, 1), (37, 129), (64, 255), (513, 2), (33, 1024), (17, 8), (250, 1008)])
def test_matrix_bit_exact(gpu, oracle, na, nb):
    rng = np.random.default_rng(na * 7 + nb)
    A = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    assert np.array_equal(gpu.hamming_matrix(A, B), oracle.hamming_matrix(A, B))


def test_best2_bit_exact_with_ties_and_mask(gpu, oracle):
    a, b, perm = synth.synth_descriptors_planted(1000)
    assert np.array_equal(gpu.hamming_best2(a, b), oracle.hamming_best2(a, b))
    # heavy ties: only 4 distinct B descriptors => first column must win
    b4 = np.repeat(b[:4], 50, axis=0)
    assert np.array_equal(gpu.hamming_best2(a[:100], b4), oracle.hamming_best2(a[:100], b4))
    rng = np.random.default_rng(9)
    mask = (rng.random((300, 1000)) < 0.05).astype(np.uint8)
    mask[5] = 0                                   # a row without candidates
    assert np.array_equal(gpu.hamming_best2(a[:300], b, mask), oracle.hamming_best2(a[:300], b, mask))


@pytest.mark.parametrize("na,nb", [(256, 1), (257, 63), (1000, 1000), (271, 130), (4096, 70)])
def test_best2_sixteen_rows_per_wave(gpu, oracle, na, nb):
    """From 256 unmasked rows on, a wavefront takes sixteen A rows at once (k_hamming_best2_rows): ragged last groups, fewer
    than two candidates, heavy ties (B drawn from 5 distinct descriptors: the first column must win)."""
    rng = np.random.default_rng(na * 13 + nb)
    A = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (5, 32), dtype=np.uint8)[rng.integers(0, 5, nb)]
    assert np.array_equal(gpu.hamming_best2(A, B), oracle.hamming_best2(A, B))
    B2 = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    assert np.array_equal(gpu.hamming_best2(A, B2), oracle.hamming_best2(A, B2))


@pytest.fixture
def mfma_forced(monkeypatch):
    """EAO_HAMMING_MFMA is read on every call: 2 = matrix cores whenever the shapes allow, 0 = popcount kernels only."""
    def set_mode(mode):
        monkeypatch.setenv("EAO_HAMMING_MFMA", str(mode))
    return set_mode


@pytest.mark.parametrize("na,nb", [(1000, 1000), (1, 8), (37, 128), (64, 256), (513, 8), (33, 1024), (129, 136), (250, 1008), (128, 128), (127, 120)])
def test_matrix_on_the_matrix_cores_bit_exact(gpu, oracle, mfma_forced, na, nb):
    """k_hamming_matrix_mfma (v_mfma_i32_32x32x32_i8 over 0 / 1 bytes, popcount(a) + popcount(b) - 2 a.b): ragged row and column blocks, the
    all-zero / all-one descriptors (distance 0 and 256), planted near-duplicates; the popcount kernels on the same input give the same matrix."""
    rng = np.random.default_rng(na * 11 + nb)
    A = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    A[0] = 0; B[0] = 255; B[-1] = 0
    if na > 3 and nb > 3: B[3] = A[3]; B[2] = A[1] ^ np.uint8(1)
    mfma_forced(2)
    got = gpu.hamming_matrix(A, B)
    assert np.array_equal(got, oracle.hamming_matrix(A, B))
    mfma_forced(0)
    assert np.array_equal(gpu.hamming_matrix(A, B), got)


@pytest.mark.parametrize("na,nb", [(1000, 1000), (1, 1), (37, 129), (129, 5), (300, 128), (128, 257), (1000, 3), (4096, 70)])
def test_best2_on_the_matrix_cores_bit_exact(gpu, oracle, mfma_forced, na, nb):
    """k_hamming_best2_mfma: keys (distance << 20 | column) straight from the accumulators; heavy ties (B drawn from 5 distinct descriptors: the first
    column must win, src/ORBmatcher.cc:102-114), fewer than two candidates, ragged blocks, the planted benchmark set."""
    rng = np.random.default_rng(na * 17 + nb)
    A = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    B5 = rng.integers(0, 256, (5, 32), dtype=np.uint8)[rng.integers(0, 5, nb)]
    B2 = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    mfma_forced(2)
    assert np.array_equal(gpu.hamming_best2(A, B5), oracle.hamming_best2(A, B5))
    assert np.array_equal(gpu.hamming_best2(A, B2), oracle.hamming_best2(A, B2))
    a, b, _ = synth.synth_descriptors_planted(min(na, 1000))
    assert np.array_equal(gpu.hamming_best2(a, b), oracle.hamming_best2(a, b))


def test_device_entry_points_with_many_pairs(gpu, oracle):
    """BASELINE configs[2] as bench.py runs it: `pairs` sets of 1000 x 1000 descriptors resident on the device, ONE launch each for the matrix and
    for best / second-best (default path selection: the matrix cores); every pair against the oracle."""
    import torch
    from eao_fusion_amd import _lib
    L = gpu.load()
    pairs, n = 9, 1000
    rng = np.random.default_rng(2002)
    A = rng.integers(0, 256, (pairs, n, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (pairs, n, 32), dtype=np.uint8)
    for p in range(pairs):      # planted matches and exact duplicates: ratio-test material and ties
        B[p, :200] = A[p, rng.permutation(n)[:200]]
        B[p, 200:260] = B[p, 0]
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    dD = torch.zeros((pairs, n, n), dtype=torch.int16, device="cuda")
    dO = torch.zeros((pairs, n, 4), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.eao_hamming_matrix_device(dA.data_ptr(), n, dB.data_ptr(), n, pairs, dD.data_ptr(), st))
    _lib.check(L.eao_hamming_best2_device(dA.data_ptr(), n, dB.data_ptr(), n, pairs, None, dO.data_ptr(), st))
    torch.cuda.synchronize()
    D, Ob = dD.cpu().numpy().view(np.uint16), dO.cpu().numpy()
    for p in range(pairs):
        assert np.array_equal(D[p], oracle.hamming_matrix(A[p], B[p])), "matrix of pair %d" % p
        assert np.array_equal(Ob[p], oracle.hamming_best2(A[p], B[p])), "best-2 of pair %d" % p


def test_descriptor_distance_known_answers(gpu):
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    m = gpu.ORBmatcher(0.6, True)
    assert m.DescriptorDistance(z, z) == 0 and m.DescriptorDistance(z, o) == 256
    assert (m.TH_LOW, m.TH_HIGH, m.HISTO_LENGTH) == (50, 100, 30)


def test_distinctive_descriptors(gpu, oracle):
    rng = np.random.default_rng(78)
    sets = []
    for n in list(rng.integers(0, 40, 300)) + [1, 64, 65, 257, 300]:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        d = np.tile(base, (int(n), 1))
        if n:
            flips = (rng.random((int(n), 256)) < rng.uniform(0.0, 0.3, (int(n), 1))).astype(np.uint8)
            d = np.packbits(np.unpackbits(d, axis=1) ^ flips, axis=1)
        sets.append(d)
    assert np.array_equal(gpu.distinctive_descriptors(sets), oracle.distinctive_descriptors(sets))
    assert len(gpu.distinctive_descriptors([])) == 0
