"""Operation lists of the recorded behaviour cases (input of
oracle/gen_golden_refcases.py; the interpreter is tests/casekit.py).

Each module defines ``CASES``: a list made with `_dsl.case`.  The lists are
written for this repository -- which behaviour of the reference a case probes
is named in its ``about`` -- and hold no reference source text.
"""
