"""CPU oracle for the StyleGAN2 generator / KD-retrain hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  This package is a from-scratch, functional (state-dict
driven) PyTorch-fp32 CPU restatement of the reference's algorithm for the hot path of SURVEY.md §8.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only
as the checker / the timed CPU baseline.  The shipped package
(`content-aware-gan-compression_amd/cagc`) never imports it.

Pinning: every function here is checked against golden vectors captured from the reference's own
CPU path by `oracle/gen_golden.py` (tests/test_oracle_golden.py).  The reference has no tests or
known-answer vectors of its own apart from the two MAC constants in
Util/Calculators.py:13-14, which tests/test_contract.py reproduces.
"""
