"""CPU oracle: a from-scratch restatement of the reference's reverse-sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this
directory; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
do, and only as the checker / the timed CPU baseline.

Pinning: the reference has no tests or golden vectors of its own,
so the oracle is pinned against outputs of the reference itself, generated in the
build container by importing /root/reference (tests/golden/make_golden.py) and
committed as small fixtures under tests/golden/.  tests/test_oracle_golden.py
checks every function here against them.
"""
