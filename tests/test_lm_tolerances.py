"""CPU suite: the LM tolerance table stays in ONE place (tests/lm_tolerances.py, VERDICT r4 next #7d) and says what it said when it was frozen.
A loosening has to edit that file -- and this test's frozen copy of it, i.e. it cannot happen by accident or in passing."""
import os
import re

import lm_tolerances as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FROZEN = dict(UPDATE_REL=1e-4, CHI2_REL=1e-6, CHI2_REL_FAR_OFF=1e-4, LAMBDA_REL=5e-4, LAMBDA_REL_FAR_OFF=4e-3, LAMBDA_REL_FAR_OFF_MAP_SCALE=2e-3, CHAOTIC_BANDS_ALLOWED=4,
              CHI2_REL_PLANES=1e-5, CHI2_REL_BATCH_VS_SINGLE=1e-4, LEAVES_THE_BAR={3039, 3040, 3045, 3059}, SCHEDULE_UNSTABLE={3050, 3055, 3059},
              CONDITIONING_BANDED_MAX=7)      # (round 6: a new entry; none of round 5's moved)


def test_the_table_is_what_was_frozen_in_round_5():
    for k, v in FROZEN.items():
        assert getattr(T, k) == v, "%s moved: %r -> %r (edit tests/lm_tolerances.py WITH its evidence, then this frozen copy)" % (k, v, getattr(T, k))
    assert len(T.CHAOTIC_BAND) == 12 and max(T.CHAOTIC_BAND.values()) == 3.7e-2 and T.LEAVES_THE_BAR <= set(T.CHAOTIC_BAND)


def test_gpu_lm_tests_carry_no_tolerance_of_their_own():
    src = open(os.path.join(ROOT, "tests", "test_gpu_lm.py")).read()
    code = "\n".join(ln.split("#")[0] for ln in src.splitlines() if not ln.lstrip().startswith(('"', "'")))
    assert not re.search(r"\b(rel|lam_rel|rtol)\s*=\s*[0-9]", code), "a literal tolerance in tests/test_gpu_lm.py: it belongs in tests/lm_tolerances.py"
    for name in ("profiles/r02_lm_trace_sensitivity.txt", "profiles/r02_lm_chaotic_seeds.txt", "profiles/r04_lm_seed3037.txt", "profiles/r05_sweeps.txt"):
        assert os.path.exists(os.path.join(ROOT, name)), "evidence file %s named by tests/lm_tolerances.py is missing" % name
