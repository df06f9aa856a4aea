"""GPU parity: DGCNN encoder + SED-Net heads (HIP EdgeConv / pointwise kernels) vs golden vectors captured
from the reference and vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def build(T, k, salt):
    from src.SEDNet import SEDNet
    from sednet_hip import synth
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
               combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=k)
    m.load_state_dict({k_: T.from_numpy(v) for k_, v in synth.closed_form_state_dict(salt).items()}, strict=True)
    return m.cuda().eval()


def test_forward_matches_reference_golden(T, golden):
    g = golden("f_e2e")
    m = build(T, int(g["k"]), int(g["salt"]))
    x = T.from_numpy(g["x"]).cuda()
    x4, feats = m.encoder(x)
    np.testing.assert_allclose(feats.cpu().numpy(), g["feats"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(x4.cpu().numpy(), g["x4"], rtol=0, atol=2e-4)
    emb, logp, loss, edges = m(x, None, False)
    assert tuple(emb.shape) == (1, 128, 512) and tuple(logp.shape) == (1, 6, 512) and tuple(edges.shape) == (1, 2, 512)
    assert tuple(loss.shape) == (1,) and float(loss) == 0.0
    np.testing.assert_allclose(emb.cpu().numpy(), g["embedding"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_prob"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(edges.cpu().numpy(), g["edges"], rtol=0, atol=5e-4)
    assert (logp.argmax(1).cpu().numpy() == g["log_prob"].argmax(1)).mean() > 0.995


@pytest.mark.parametrize("k,N,B", [(20, 700, 2), (64, 333, 1)])
def test_forward_matches_oracle(T, k, N, B):
    """batched clouds, ragged N (not a multiple of 32/128), reference-default k = 64."""
    from oracle import backbone
    from sednet_hip import synth
    x, _, _ = synth.batch_clouds(B, N, seed0=50)
    params = synth.closed_form_state_dict(2)
    m = build(T, k, 2)
    emb, logp, _, edges = m(T.from_numpy(x).cuda(), None, False)
    oe, ol, oed = backbone.sednet_forward(params, x, k)
    np.testing.assert_allclose(emb.cpu().numpy(), oe, rtol=0, atol=5e-4)
    np.testing.assert_allclose(logp.cpu().numpy(), ol, rtol=0, atol=5e-4)
    np.testing.assert_allclose(edges.cpu().numpy(), oed, rtol=0, atol=5e-4)


def test_batch_invariance(T):
    """a cloud's outputs do not depend on what else is in the batch (GroupNorm is per sample)."""
    from sednet_hip import synth
    x, _, _ = synth.batch_clouds(3, 512, seed0=80)
    m = build(T, 20, 1)
    xb = T.from_numpy(x).cuda()
    eb = m(xb)[0]
    e1 = m(xb[1:2])[0]
    np.testing.assert_array_equal(eb[1].cpu().numpy(), e1[0].cpu().numpy())


def test_weight_cache_follows_in_place_updates(T):
    """the kernel-layout weight cache is invalidated when parameters change in place (optimizer steps, manual edits)."""
    from sednet_hip import synth
    x, _, _ = synth.batch_clouds(1, 300, seed0=5)
    m = build(T, 20, 1)
    xb = T.from_numpy(x).cuda()
    e0 = m(xb)[0].clone()
    with T.no_grad():
        m.mlp_seg_prob2.weight.mul_(2.0)
        m.mlp_seg_prob2.bias.mul_(2.0)
    e1 = m(xb)[0]
    np.testing.assert_allclose(e1.cpu().numpy(), 2.0 * e0.cpu().numpy(), rtol=1e-5, atol=1e-6)
