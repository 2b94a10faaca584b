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
