"""CPU oracle for the DMCF hot path -- TEST INFRASTRUCTURE ONLY (parity unpinned).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing in ``dmcf_amd`` (the product) imports it; ``tests/test_abi.py::test_product_does_not_import_oracle`` checks that.

See ``oracle/dmcf_oracle.c`` for what is restated and why parity against TensorFlow/Open3D is
"unpinned" (the reference ships no golden vectors and its arithmetic lives in open3d==0.15.2).
"""
from .oracle import *  # noqa: F401,F403
