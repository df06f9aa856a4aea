"""CPU: HPNet spectral step -- the oracle against golden data captured from the
reference (the device path is checked against the same data in tests/test_gpu_hpnet.py)."""
import numpy as np

from oracle import hpnet as ohp


def test_oracle_affinity_and_entropy_match_reference(golden):
    g = golden("f_hpnet")
    A, nnid = ohp.affinity_matrix_normal(g["p"], g["n"])
    same = (nnid[:40] == g["nnid"]).mean()
    assert same > 0.98                                      # farthest-50 selection (near-ties aside)
    np.testing.assert_allclose(A[:40], g["A_rows"], rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(A.sum(1), g["A_rowsum"], rtol=2e-4)
    np.testing.assert_allclose(ohp.compute_entropy(g["feat"][0], int(g["chunk"])), g["ent_feat"], rtol=1e-4)
    np.testing.assert_allclose(ohp.compute_entropy(g["v"][0], int(g["chunk"])), g["ent_v"], rtol=1e-4)
