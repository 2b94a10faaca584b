"""Frozen vectors (tests/golden/*.npz, written by tools/gen_golden.py from the CPU oracle) against BOTH implementations: the
oracle of the current tree (CPU suite) and the HIP path (-m gpu).  See tests/golden_cases.py for what each case holds and why
(oracle and kernels were only ever compared with each other at head)."""
import os

import numpy as np
import pytest

import golden_cases as GC

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    path = os.path.join(GOLDEN, name + ".npz")
    assert os.path.exists(path), "missing fixture %s: run tools/gen_golden.py %s" % (path, name)
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def test_every_case_has_a_fixture_and_nothing_else_lies_there():
    have = {f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz")}
    assert have == set(GC.CASES), sorted(have ^ set(GC.CASES))


@pytest.mark.parametrize("name", list(GC.CASES))
def test_oracle_matches_golden(oracle, name):
    """The oracle must reproduce its own frozen outputs: integer tables bit for bit; the fp64 LM results are deterministic code on
    IEEE arithmetic (-ffp-contract=off) and are held to 1e-9 of the update (libm differences between hosts only)."""
    got = GC.CASES[name](GC.OracleApi(oracle))
    GC.compare(name, got, _load(name), lm_rel=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GC.CASES))
def test_gpu_matches_golden(name):
    import torch  # noqa: F401
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    got = GC.CASES[name](GC.ProductApi(E))
    GC.compare(name, got, _load(name))
