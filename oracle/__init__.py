"""CPU oracle of the hot path — TEST INFRASTRUCTURE ONLY.

Plain numpy / pure-python restatements of the reference algorithms (each function cites the
reference file:line it follows).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this package, and only as the checker: the product
(`pipelinerl_amd/`) never imports it and has no CPU fallback.

Form: the reference is pure Python (no C/C++ sources to compile into `oracle/_ref`), so the
restatement is numpy / pure Python rather than C, and `__graft_entry__.build()` has no oracle
binary to build (it builds the test-only host harness of the device token math instead).

Pinning: the reference ships no tests or golden vectors for this path, so
the oracle is pinned against outputs of the reference's own functions run in the build
container: `tests/golden/make_golden.py` imports `/root/reference/pipelinerl` and writes the
fixtures under `tests/golden/`; `tests/test_oracle_golden.py` checks every oracle function
against them.  The micro-batch schedule loop is cut from the reference source and executed with
recording stubs; the `files` stream backend is executed with stand-ins for its two missing imports;
the weight-update sender / receiver, the trainer messages and TrainerState are cut from the source and
executed with recording stubs for their absent dependencies; see DESIGN.md "parity status".
"""
