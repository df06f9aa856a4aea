"""CPU: HPNet spectral step -- oracle and the torch restatement against golden data
captured from the reference (deterministic parts exactly; the lobpcg eigenvectors statistically)."""
import numpy as np
import torch

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


def test_mirror_matches_reference(golden):
    from src import smooth_normal_matrix as snm
    g = golden("f_hpnet")
    P, Nn, Ft = torch.from_numpy(g["p"][None]), torch.from_numpy(g["n"][None]), torch.from_numpy(g["feat"])
    CH = int(g["chunk"])
    A = snm.construction_affinity_matrix_normal(P, Nn, sigma=0.1, knn=50)
    np.testing.assert_allclose(A[0, :40].numpy(), g["A_rows"], rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(float(snm.compute_entropy(Ft, CHUNK=CH)), g["ent_feat"], rtol=1e-4)
    np.testing.assert_allclose(float(snm.compute_entropy(torch.from_numpy(g["v"]), CHUNK=CH)), g["ent_v"], rtol=1e-4)
    torch.manual_seed(7)
    out = snm.hpnet_process(Ft, P, Nn, normal_smooth_w=0.5, CHUNK=CH).numpy()
    assert out.shape == g["out"].shape == (1, 600, 28)
    # the embedding block is deterministic: feat * (1.7 - entropy)
    np.testing.assert_allclose(out[..., :16], g["out"][..., :16], rtol=1e-4, atol=1e-6)
    # the spectral block: unit rows, and the 12-d subspace agrees with the reference's (lobpcg is an
    # unconverged random-start iteration: compare through principal angles)
    v, vr = out[0, :, 16:], g["out"][0, :, 16:]
    qa, _ = np.linalg.qr(v)
    qb, _ = np.linalg.qr(vr)
    sv = np.linalg.svd(qa.T @ qb, compute_uv=False)
    assert sv.min() > 0.9, sv
