"""ORBmatcher -- the distance kernel of the reference matcher (reference include/ORBmatcher.h:37-102,
src/ORBmatcher.cc:37-39,1649-1665).  Distances come from libeaofusion_hip.so; the greedy assignment passes of
the Search* routines replay on the host over these distances."""
import numpy as np

from . import _lib


def hamming_matrix(A, B):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    D = np.zeros((len(A), len(B)), np.uint16)
    _lib.check(_lib.load().eao_hamming_matrix(_lib.ptr(A), len(A), _lib.ptr(B), len(B), _lib.ptr(D)))
    return D


def hamming_best2(A, B, mask=None):
    """Per row of A: (best, second, idx, idx2) over the allowed columns; first column wins ties."""
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    out = np.zeros((len(A), 4), np.int32)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
        assert mask.shape == (len(A), len(B))
    _lib.check(_lib.load().eao_hamming_best2(_lib.ptr(A), len(A), _lib.ptr(B), len(B), _lib.ptr(mask), _lib.ptr(out)))
    return out


class ORBmatcher:
    TH_LOW = 50
    TH_HIGH = 100
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    @staticmethod
    def DescriptorDistance(a, b):
        return int(hamming_matrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32))[0, 0])
