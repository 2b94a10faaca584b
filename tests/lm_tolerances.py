"""THE tolerance table of the LM parity tests (VERDICT r4 next #7d): every bound the GPU's LocalBundleAdjustment / BundleAdjustment / PoseOptimization results are held
to against oracle/lm_cpu.cpp lives here, each with the evidence file that justifies it.  A future loosening edits THIS file and nothing else -- tests/test_gpu_lm.py
imports every number below; tests/test_lm_tolerances.py (CPU suite) fails when test_gpu_lm.py grows a literal tolerance of its own again.

north_star: "within 1e-4 relative on BA pose / point updates".  UPDATE_REL is that figure and has never moved.  The other entries bound the LM TRACE (lambda, chi2 per
iteration), which north_star does not name and which amplifies rounding: lambda is multiplied by a cubic of rho = a ratio of small differences in every iteration."""

# |GPU - oracle| <= UPDATE_REL * max |oracle - input| + 2 float32 ulps of the value: poses and points, every LM test.  Unchanged since round 1.
UPDATE_REL = 1e-4

# ---- the LM trace, compared on the well-conditioned prefix of every optimize() call (until chi2 stalls at the float32 noise floor)
# chi2 per iteration, relative
CHI2_REL = 1e-6              # test_local_ba_parity & co.  Observed 8e-10 (six of eight problems 1e-15); one float32 ulp on the inputs moves the ORACLE's own chi2 by
                             # up to 3e-6 (profiles/r02_lm_trace_sensitivity.txt)
CHI2_REL_FAR_OFF = 1e-4      # the far-off-start families (25 degrees, 0.8 m off; monocular-heavy): round 1
# lambda per iteration, relative
LAMBDA_REL = 5e-4            # round 2 (2e-3 in round 1): observed 5.1e-5; the oracle's own one-ulp band is 8e-3 (profiles/r02_lm_trace_sensitivity.txt)
LAMBDA_REL_FAR_OFF = 4e-3    # round 4 (2e-3 before), the tile-solver path: the pair assembly works on Cholesky-scaled blocks W = Hpl C^-T (csrc/lm.hip, ba_chol3) where the
                             # oracle multiplies by an explicit 3 x 3 inverse as upstream does -- different rounding from the first iteration on.  Seed 3037, iteration 14:
                             # 3.07e-3 off (1.45e-3 with EAO_BA_WMODE=0) while chi2 agrees to 9e-8 and the points to 8e-6 of the update (profiles/r04_lm_seed3037.txt)
LAMBDA_REL_FAR_OFF_MAP_SCALE = 2e-3      # the same windows on the map-scale path (explicit inverse there): round 1's figure, never moved

# ---- far-off seeds 3030..3059 on which the ORACLE ITSELF moves by more than UPDATE_REL -- or changes its LM schedule -- when its inputs are perturbed by ONE float32
#      ulp: seed -> the larger of its pose / point displacement relative to the update (profiles/r02_lm_chaotic_seeds.txt, written by tools/lm_chaotic_seeds.py, "B"
#      columns).  No two implementations -- not even two summation orders of one -- can be held to 1e-4 there.
CHAOTIC_BAND = {3031: 2.2e-4, 3034: 1.8e-4, 3039: 1.7e-3, 3040: 3.7e-2, 3042: 1.9e-4, 3045: 1.2e-3, 3048: 2.5e-4, 3050: 1e-4, 3051: 2.8e-4,
                3053: 4.7e-3, 3055: 1.2e-4, 3059: 1e-4}
# ... of which the oracle's own LM schedule (iterations of the two optimize() calls, outlier table) changes under that perturbation:
SCHEDULE_UNSTABLE = {3050, 3055, 3059}
# ... and on which the GPU result actually leaves UPDATE_REL ("G" columns of the same log; every other banded seed is still held to UPDATE_REL):
LEAVES_THE_BAR = {3039, 3040, 3045, 3059}
# the update bound on LEAVES_THE_BAR seeds: this many of the oracle's own one-ulp bands
CHAOTIC_BANDS_ALLOWED = 4

# ---- plane landmarks (g2o's central-difference Jacobians, delta = 1e-9: ~1e-7 of rounding noise in ANY implementation), BundleAdjustment with map planes: round 2
CHI2_REL_PLANES = 1e-5
# ---- the batched entry point against single calls of the same windows: chi2 after each optimize(), relative (the batch adds a landmark's terms in another order)
CHI2_REL_BATCH_VS_SINGLE = 1e-4

# ---- round 6: tests/test_gpu_lm_conditioning.py -- the five excursions of the round-5 sweeps (profiles/r05_sweeps.txt: one weakly constrained landmark each,
#      1.1 - 1.9e-4 of the update) + 50 fresh draws.  A problem outside UPDATE_REL must lie inside the oracle's own one-ulp band, measured in the test; at most this many may.
CONDITIONING_BANDED_MAX = 7
