"""CPU oracle for the AutoGnothi masked-forward / Shapley hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy (fp32) restatement of the reference
algorithms (each function cites the reference file:line it follows).  It may be imported only
by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` — as the
checker or the reported CPU baseline, never as the thing measured or shipped.  Nothing under
``autognothi_amd/`` imports it; the product path raises if the HIP library is missing.

Pinning: the reference has no golden vectors or known-answer tests for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself, generated in
the build container by ``tests/golden/make_golden.py`` (imports /root/reference) and committed
as ``tests/golden/*.npz``; ``tests/test_oracle_*.py`` check every function here against them
(masks/indices bit-exact, fp32 values <= 1e-5 abs, the reference's own tolerance at
scripts/train_all.py:212-215).
"""
