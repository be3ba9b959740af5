"""CPU restatement (PyTorch-CPU, fp32) of RobIR's per-ray forward renderer hot path.

TEST INFRASTRUCTURE.  This package is the parity *oracle*: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it, and there only as the checker / the timed CPU
baseline -- never as part of the product path (robir_amd/ never imports it).

Every function cites the reference file:line whose arithmetic it restates
(paths relative to ingra14m/RobIR).  The restatement is functional (plain dict of weights,
explicit RNG draws as arguments) rather than the reference's nn.Module tree.

Pinning: the reference has no tests or golden vectors of its own, so the
oracle is pinned against outputs of the reference itself, recorded in this container by
oracle/gen_golden.py into tests/golden/*.npz, and re-checked by tests/test_oracle_golden.py.
"""
from . import encoding, nets, neus, octree, sg, renderer, raytracing  # noqa: F401
