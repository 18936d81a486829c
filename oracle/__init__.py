"""CPU oracle for the pose hot path -- TEST INFRASTRUCTURE ONLY.

A plain numpy / PyTorch-CPU restatement of the reference's algorithms for every row of
SURVEY.md §8(a).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this package, and only as the checker -- never as the thing measured or
shipped.  The product package (`infantposeestimation_gaussianbias_amd`) must not import it.

Parity status: PINNED.  Every function here is checked in `tests/test_oracle_golden.py`
against vectors captured from the reference's own modules imported in the build container
(`tests/golden/make_golden.py`, fixtures under `tests/golden/`).  The reference ships no
tests or golden vectors of its own.

Modules: target (T1,T2) · decode (D1-D4) · losses (L1-L4) · nets (A1-A7,H1,H2) · optim (S1).
"""
