"""CPU oracle for the InvertAvatar generator forward pass.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain torch-CPU / numpy, the
arithmetic of the reference's generator hot path so that the
HIP backend in ``invertavatar_amd`` can be checked against it on a box that has no
copy of the reference.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never does.

Pinning: the reference ships no tests or golden vectors for this path (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself, generated
in the build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference`` on CPU) and committed under ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every oracle stage against those fixtures.

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
