"""CPU oracle for the t2v-turbo denoise hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product path
(``t2v-turbo_amd/``) never imports this package and fails loudly when the HIP
library is missing.

Every function is a plain fp32 torch-CPU restatement of the reference algorithm
and cites the reference file:line it follows.  The restatement is pinned against
golden vectors produced by importing the reference itself in the build
container (``tests/golden/make_golden.py``; fixtures committed under
``tests/golden/``).  Parts of the path whose arithmetic lives in un-vendored
third-party code (ModelScope backbone -> diffusers 0.30.0) are marked
"parity unpinned" where they appear.
"""
