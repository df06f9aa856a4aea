"""CPU oracle for the SED-Net hot path (TEST INFRASTRUCTURE -- not product code).

A plain numpy restatement of the reference algorithm for the path named by
BASELINE.json's north_star (DGCNN backbone -> mean-shift -> primitive LSQ fits),
each function citing the /root/reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package, and only as the checker / the timed CPU baseline. The
product path (`sed-net_amd/`) never imports it and fails loudly when the HIP
extension is missing.

Parity pin: the reference ships no golden vectors or asserting tests for this
path, so the oracle is pinned against outputs of the
reference itself, captured in the build container by
`tests/golden/make_golden.py` (reference imported under `tests/golden/ref_shim.py`)
and committed as `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every
oracle function against them.
"""
