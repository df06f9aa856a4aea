"""GPU parity of the training step: fused HIP forward + HIP backward (sednet_hip/autograd.py,
csrc/edgeconv_bwd.hip) against torch.autograd on the CPU restatement (oracle/train.py) and against gradients captured
from the reference model itself (tests/golden/f_train.npz). Tolerances: gradients are sums of O(N k) fp32 terms
accumulated in a different order (and scattered with atomics), so they are compared relative to the largest entry."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "needs the MI355X"
    return torch


def close(got, ref, rel, msg=""):
    got, ref = np.asarray(got), np.asarray(ref)
    scale = max(float(np.abs(ref).max()), 1e-12)
    np.testing.assert_allclose(got, ref, rtol=0, atol=rel * scale, err_msg=msg)


@pytest.mark.parametrize("C,Cout,N,k,B", [(64, 64, 300, 12, 2), (64, 128, 257, 20, 1), (6, 64, 200, 9, 2), (64, 128, 150, 64, 1),
                                          (6, 64, 130, 64, 1)])
def test_edgeconv_backward_matches_autograd(T, C, Cout, N, k, B):
    import torch.nn.functional as F
    from oracle import train as ot
    from sednet_hip import autograd as hag
    rng = np.random.default_rng(C + Cout + N)
    x = rng.normal(size=(B, C, N)).astype(np.float32)
    W = (rng.normal(size=(Cout, 2 * C, 1, 1)) / np.sqrt(2 * C)).astype(np.float32)
    gamma = rng.normal(size=Cout).astype(np.float32)               # both signs: max and min selection
    beta = rng.normal(size=Cout).astype(np.float32)
    cot = rng.normal(size=(B, Cout, N)).astype(np.float32)
    idx = np.stack([np.stack([rng.permutation(N)[:k] for _ in range(N)]) for _ in range(B)]).astype(np.int64)

    xc, Wc, gc, bc = (T.from_numpy(a).clone().requires_grad_(True) for a in (x, W, gamma, beta))
    out_c = ot._edge_block(xc, T.from_numpy(idx), Wc, gc, bc, 2)
    (out_c * T.from_numpy(cot)).sum().backward()

    ld = 8 if C == 6 else C
    xp = np.zeros((B, N, ld), np.float32)
    xp[:, :, :C] = x.transpose(0, 2, 1)
    xg = T.from_numpy(xp).cuda().requires_grad_(C != 6)
    Wg, gg, bg = (T.from_numpy(a).cuda().requires_grad_(True) for a in (W, gamma, beta))
    out_g = hag.EdgeConvGN.apply(xg, T.from_numpy(idx.astype(np.int32)).cuda(), Wg, gg, bg, C, 2, 1e-5, 0.2)
    (out_g * T.from_numpy(cot.transpose(0, 2, 1).copy()).cuda()).sum().backward()

    close(out_g.detach().cpu().numpy().transpose(0, 2, 1), out_c.detach().numpy(), 2e-5, "forward")
    close(Wg.grad.cpu().numpy(), Wc.grad.numpy(), 2e-4, "dW")
    close(gg.grad.cpu().numpy(), gc.grad.numpy(), 2e-4, "dgamma")
    close(bg.grad.cpu().numpy(), bc.grad.numpy(), 2e-4, "dbeta")
    if C != 6:
        close(xg.grad.cpu().numpy()[:, :, :C].transpose(0, 2, 1), xc.grad.numpy(), 2e-4, "dx")


@pytest.mark.parametrize("Cout,N,k,B", [(64, 1000, 20, 3), (128, 777, 64, 2)])
def test_edgeconv_input_gradient_is_deterministic(T, Cout, N, k, B):
    """Default backward (ops.DETERMINISTIC_BWD): per-edge contributions stored and gathered per target row in ascending edge
    order through the reverse graph -- the same bits on every run, equal to the fp32-atomic path up to summation order; the
    reverse graph lists, per target, exactly its incoming edges in ascending order (hub rows with hundreds of them)."""
    from sednet_hip import ops
    g = T.Generator().manual_seed(N + k)
    C = 64
    x = T.randn(B, N, C, generator=g).cuda()
    # a hubby graph: half of every point's neighbours come from the first 20 rows
    idx = T.randint(0, N, (B, N, k), generator=g)
    idx[:, :, ::2] = T.randint(0, 20, (B, N, (k + 1) // 2), generator=g)
    idx = idx.int().cuda()
    rptr, redge = ops.reverse_graph(idx)
    flat = idx.reshape(B, N * k).cpu().numpy()
    rp, re = rptr.cpu().numpy(), redge.cpu().numpy()
    assert (rp[:, 0] == 0).all() and (rp[:, -1] == N * k).all()
    for b in range(B):
        for t in (0, 7, 19, 20, N // 2, N - 1):
            mine = re[b, rp[b, t]:rp[b, t + 1]]
            np.testing.assert_array_equal(mine, np.nonzero(flat[b] == t)[0])          # all of them, ascending
    assert int((rp[:, 1:21] - rp[:, 0:20]).max()) > 10 * k
    W1t = (T.randn(C, Cout, generator=g) / 8).cuda()
    W2t = (T.randn(C, Cout, generator=g) / 8).cuda()
    S = T.randn(B, N, Cout, generator=g).cuda()
    jsel = T.randint(0, k, (B, N, Cout), generator=g).to(T.uint8).cuda()
    ak = (T.randn(B, 2, 2, generator=g) * 0.1).cuda()
    runs = [ops.edgeconv_bwd(x, C, idx, W1t, W2t, 2, S, jsel, ak, True, deterministic=True) for _ in range(3)]
    for r in runs[1:]:
        for a, b_ in zip(r, runs[0]):
            assert T.equal(a, b_)                                                     # bit-identical run to run
    atom = ops.edgeconv_bwd(x, C, idx, W1t, W2t, 2, S, jsel, ak, True, deterministic=False)
    assert T.equal(atom[0], runs[0][0]) and T.equal(atom[1], runs[0][1])              # weight gradients: same kernel
    scale = float(atom[2].abs().max())
    np.testing.assert_allclose(runs[0][2].cpu().numpy(), atom[2].cpu().numpy(), atol=2e-5 * scale)
    # bf16 products (edgeconv_bwd_input_bf16_kernel: all slabs in one workgroup, one E slab): reproducible, and the fp32
    # result up to bf16 rounding of W, x_j - x_p and dy (relative 2^-9 each, averaged over 64 / Cout terms)
    b16 = [ops.edgeconv_bwd(x, C, idx, W1t, W2t, 2, S, jsel, ak, True, deterministic=True, bf16=True) for _ in range(2)]
    assert T.equal(b16[0][2], b16[1][2])
    err = (b16[0][2] - runs[0][2]).abs()
    assert float(err.max()) < 2e-2 * scale and float(err.mean()) < 2e-3 * scale, (float(err.max()) / scale, float(err.mean()) / scale)
    assert T.equal(b16[0][2][:, :, C:], T.zeros_like(b16[0][2][:, :, C:]))
    for a, b_, f in zip(b16[0][:2], b16[1][:2], runs[0][:2]):                          # weight gradients (bf16 kernel)
        assert T.equal(a, b_)
        e = (a - f).abs()
        sc = float(f.abs().max())
        assert float(e.max()) < 2e-2 * sc and float(e.mean()) < 3e-3 * sc, (float(e.max()) / sc, float(e.mean()) / sc)


@pytest.mark.parametrize("K,Cout,G,act,N,B,cb", [(256, 512, 8, 1, 333, 2, True), (256, 128, 4, 0, 200, 1, False),
                                                   (512, 256, 4, 1, 130, 3, False)])
def test_pointwise_backward_matches_autograd(T, K, Cout, G, act, N, B, cb):
    import torch.nn.functional as F
    from sednet_hip import autograd as hag
    rng = np.random.default_rng(K + Cout + N)
    X = rng.normal(size=(B, N, K)).astype(np.float32)
    W = (rng.normal(size=(Cout, K, 1)) / np.sqrt(K)).astype(np.float32)
    bias = rng.normal(size=Cout).astype(np.float32)
    cbias = rng.normal(size=(B, Cout)).astype(np.float32)
    gamma, beta = rng.normal(size=Cout).astype(np.float32), rng.normal(size=Cout).astype(np.float32)
    cot = rng.normal(size=(B, N, Cout)).astype(np.float32)

    def run(dev):
        t = lambda a: T.from_numpy(a).to(dev).requires_grad_(True)
        Xt, Wt, bt, cbt, gt, bet = t(X), t(W), t(bias), t(cbias), t(gamma), t(beta)
        if dev == "cpu":
            y = F.conv1d(Xt.transpose(1, 2), Wt, bt)
            if cb:
                y = y + cbt[:, :, None]
            y = F.group_norm(y, G, gt, bet, 1e-5)
            out = (F.relu(y) if act == 1 else y).transpose(1, 2)
        else:
            out = hag.ConvGNAct.apply(Xt, Wt, bt, cbt if cb else None, gt, bet, G, 1e-5, act)
        (out * T.from_numpy(cot).to(dev)).sum().backward()
        grads = [Xt.grad, Wt.grad, bt.grad, gt.grad, bet.grad] + ([cbt.grad] if cb else [])
        return out.detach().cpu().numpy(), [g.cpu().numpy() for g in grads]

    oc, gc = run("cpu")
    og, gg = run("cuda")
    close(og, oc, 2e-5, "forward")
    for name, a, b in zip(["dX", "dW", "dbias", "dgamma", "dbeta", "dcbias"], gg, gc):
        close(a, b, 2e-4, name)


def _model(T, k, salt):
    from sednet_hip import synth
    from src.SEDNet import SEDNet
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
               combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=k)
    m.load_state_dict({n: T.from_numpy(v) for n, v in synth.closed_form_state_dict(salt).items()})
    return m.cuda().train()


def _check_digest(g, prefix, named_grads, rel):
    from train_case import grad_digest
    dig = grad_digest(named_grads)
    names = [key[len(prefix) + 2:] for key in g.files if key.startswith(prefix + "g:")]
    assert len(names) > 40
    for n in names:
        close(dig["g:" + n], g[prefix + "g:" + n], rel, n)
        np.testing.assert_allclose(dig["n:" + n][1], g[prefix + "n:" + n][1], rtol=20 * rel, err_msg=n)


def _mostly_close(got, ref, rel, frac=0.995):
    """A near-tie at the k-th neighbour can resolve differently on the device (torch.topk's tie order is unspecified);
    the handful of points it touches are allowed to differ."""
    got, ref = np.asarray(got), np.asarray(ref)
    ok = np.abs(got - ref) <= rel * max(float(np.abs(ref).max()), 1e-12)
    assert ok.mean() >= frac, f"{(1 - ok.mean()) * 100:.3f} % of the entries differ"


def test_model_gradients_match_reference(T, golden):
    """Linear functional of the three outputs, d/dparams on the HIP path:
    (a) against torch.autograd on the CPU restatement fed with the device's own neighbour graphs (tight);
    (b) against the reference's own gradients (golden; loose only because a few near-tie neighbours may differ)."""
    from oracle import train as ot
    from sednet_hip import synth
    from train_case import train_case
    g = golden("f_train")
    x, labels, types, edges, edges_w, cot = train_case(synth, int(g["N"]), int(g["B"]))
    m = _model(T, int(g["k"]), int(g["salt"]))
    emb, logp, _, ed = m(T.from_numpy(x).cuda())
    c = {n: T.from_numpy(v).cuda() for n, v in cot.items()}
    L = (emb * c["emb"]).sum() + (logp * c["logp"]).sum() + (ed * c["edges"]).sum()
    L.backward()
    grads = {n: p.grad.cpu().numpy() for n, p in m.named_parameters() if p.grad is not None}

    # (a) oracle with the same graphs
    idx = tuple(i.cpu().long() for i in m.encoder.last_graphs)
    p = {n: T.from_numpy(v).clone().requires_grad_(True) for n, v in synth.closed_form_state_dict(int(g["salt"])).items()
         if v.dtype == np.float32 and not n.startswith("pos_enc")}
    emb_o, logp_o, ed_o, _ = ot.sednet_forward(p, T.from_numpy(x), int(g["k"]), idx=idx)
    close(emb.detach().cpu().numpy(), emb_o.detach().numpy(), 2e-5, "embedding")
    close(logp.detach().cpu().numpy(), logp_o.detach().numpy(), 2e-5, "log_prob")
    close(ed.detach().cpu().numpy(), ed_o.detach().numpy(), 2e-5, "edges")
    Lo = (emb_o * T.from_numpy(cot["emb"])).sum() + (logp_o * T.from_numpy(cot["logp"])).sum() \
        + (ed_o * T.from_numpy(cot["edges"])).sum()
    Lo.backward()
    assert len(grads) > 40
    for n, v in p.items():
        if v.grad is not None:
            close(grads[n], v.grad.numpy(), 5e-4, n)

    # (b) the reference itself
    _mostly_close(emb.detach().cpu().numpy(), g["emb"], 2e-5)
    _mostly_close(logp.detach().cpu().numpy(), g["logp"], 2e-5)
    _mostly_close(ed.detach().cpu().numpy(), g["edges_pred"], 2e-5)
    np.testing.assert_allclose(L.item(), float(g["lin_loss"]), rtol=2e-3)
    _check_digest(g, "lin/", [(n, T.from_numpy(v)) for n, v in grads.items()], 2e-2)


def test_training_loss_matches_reference(T, golden):
    """train_sed_net.py:250-271 on the device (losses of src/segment_loss.py, src/My_edge_loss.py) with np.random seeded
    like the fixture: loss terms and the gradient of the total."""
    from sednet_hip import synth
    from sednet_hip.train import training_loss
    from train_case import train_case
    g = golden("f_train")
    x, labels, types, edges, edges_w, cot = train_case(synth, int(g["N"]), int(g["B"]))
    m = _model(T, int(g["k"]), int(g["salt"]))
    np.random.seed(11)
    loss, parts = training_loss(m, T.from_numpy(x).cuda(), T.from_numpy(labels).cuda(), T.from_numpy(types).cuda(),
                                T.from_numpy(edges).cuda(), T.from_numpy(edges_w).cuda(), smoothing=0.025)
    loss.backward()
    got = np.array([loss.item(), parts["embed"], parts["type"], parts["edge"], parts["edge_embed"]])
    np.testing.assert_allclose(got, g["loss"], rtol=2e-3)
    _check_digest(g, "loss/", [(n, p.grad.cpu()) for n, p in m.named_parameters() if p.grad is not None], 2e-2)


def test_training_gradients_are_bit_reproducible(T):
    """The whole step -- fused forwards, HIP backward with the reverse-graph gather, own split-K GEMM, fixed-order
    reductions, the losses' sort-based index backward -- twice from the same state: identical loss and gradients, bit for bit
    (with the fp32-atomic scatter, ops.DETERMINISTIC_BWD = False, runs differ in the last bits)."""
    from sednet_hip import synth
    from sednet_hip.train import training_loss
    from train_case import train_case
    x, labels, types, edges, edges_w, _ = train_case(synth, 3000, 2, seed0=321)
    batch = tuple(T.from_numpy(a).cuda() for a in (x, labels, types, edges, edges_w))
    runs = []
    for _ in range(3):
        m = _model(T, 20, 3)
        np.random.seed(7)
        loss, _ = training_loss(m, *batch, smoothing=0.025)
        loss.backward()
        runs.append((loss.item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    assert len(runs[0][1]) > 40
    for l, gr in runs[1:]:
        assert l == runs[0][0]
        for n, v in gr.items():
            assert T.equal(v, runs[0][1][n]), n


def test_training_steps_reduce_the_loss(T):
    """A few AdamW steps on one synthetic batch at 2 x 2048 points, k = 20: the loss goes down and stays finite."""
    from sednet_hip import synth
    from sednet_hip.train import train_step
    from train_case import train_case
    x, labels, types, edges, edges_w, _ = train_case(synth, 2048, 2, seed0=500)
    m = _model(T, 20, 3)
    opt = T.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0)
    batch = tuple(T.from_numpy(a).cuda() for a in (x, labels, types, edges, edges_w))
    hist = []
    for it in range(6):
        np.random.seed(5)                                        # same triplets every step: a deterministic objective
        hist.append(train_step(m, opt, batch, smoothing=0.025)["loss"])
    assert np.isfinite(hist).all()
    assert hist[-1] < hist[0], hist


def _bf16_round(T, a):
    return a.to(T.bfloat16).float()


@pytest.mark.parametrize("M,N,K,tA,tB", [(300, 256, 512, False, False), (256, 512, 40000, True, False),
                                         (130, 70, 1000, False, True), (64, 128, 33, True, True)])
def test_own_gemm_matches_torch(T, M, N, K, tA, tB):
    """gemm.hip (the backward products of the pointwise layers; round 1 called rocBLAS): fp32 mode against a float64
    product, bf16 mode against the float64 product of the bf16-rounded operands (the only difference then is the fp32
    accumulation order); split-K (the 40 000-row reduction) is deterministic."""
    from sednet_hip import ops
    g = T.Generator().manual_seed(M + N + K)
    A = T.randn((K, M) if tA else (M, K), generator=g).cuda()
    B = T.randn((N, K) if tB else (K, N), generator=g).cuda()
    if K % 4 or (M % 4 and tA) or (N % 4 and not tB):                   # row strides must be multiples of 4 floats
        A = T.nn.functional.pad(A, (0, (-A.shape[1]) % 4))[:, :A.shape[1]]
        B = T.nn.functional.pad(B, (0, (-B.shape[1]) % 4))[:, :B.shape[1]]
    opA = (lambda t: t.t()) if tA else (lambda t: t)
    opB = (lambda t: t.t()) if tB else (lambda t: t)
    ref = (opA(A).double() @ opB(B).double()).cpu().numpy()
    got = ops.gemm(A, B, tA, tB, bf16=False)
    close(got.cpu().numpy(), ref, 3e-6 * np.sqrt(K), "fp32")
    assert T.equal(got, ops.gemm(A, B, tA, tB, bf16=False))               # fixed-order split-K
    refb = (opA(_bf16_round(T, A)).double() @ opB(_bf16_round(T, B)).double()).cpu().numpy()
    close(ops.gemm(A, B, tA, tB, bf16=True).cpu().numpy(), refb, 3e-6 * np.sqrt(K), "bf16")


def test_bf16_layers_equal_fp32_layers_on_rounded_operands(T):
    """bf16 products = fp32 products of the bf16-rounded operands (fp32 accumulate): pointwise layer and the 64-channel
    EdgeConv layer, forward values and all gradients (the EdgeConv backward kernels stay fp32)."""
    from sednet_hip import autograd as hag, ops
    rng = np.random.default_rng(0)
    B, N, K, Cout, G = 2, 300, 256, 128, 4
    X = T.from_numpy(rng.normal(size=(B, N, K)).astype(np.float32)).cuda()
    W = T.from_numpy((rng.normal(size=(Cout, K, 1)) / 16).astype(np.float32)).cuda()
    bias, gamma, beta = (T.from_numpy(rng.normal(size=Cout).astype(np.float32)).cuda() for _ in range(3))
    cot = T.from_numpy(rng.normal(size=(B, N, Cout)).astype(np.float32)).cuda()

    def run(Xin, Win, bf16):
        ops.TRAIN_BF16 = bf16
        try:
            Xt, Wt = Xin.clone().requires_grad_(True), Win.clone().requires_grad_(True)
            out = hag.ConvGNAct.apply(Xt, Wt, bias, None, gamma, beta, G, 1e-5, 1)
            (out * cot).sum().backward()
            return out.detach(), Xt.grad, Wt.grad
        finally:
            ops.TRAIN_BF16 = False
    ob, dXb, dWb = run(X, W, True)
    of, _, _ = run(_bf16_round(T, X), _bf16_round(T, W), False)
    close(ob.cpu().numpy(), of.cpu().numpy(), 2e-5, "pointwise forward")
    o32, dX32, dW32 = run(X, W, False)
    close(ob.cpu().numpy(), o32.cpu().numpy(), 2e-2, "pointwise bf16 vs fp32")
    _mostly_close(dXb.cpu().numpy(), dX32.cpu().numpy(), 3e-2, 0.99)       # a ReLU unit near 0 may switch after rounding
    _mostly_close(dWb.cpu().numpy(), dW32.cpu().numpy(), 3e-2, 0.999)
    # EdgeConv, C = 64
    C, Co, k = 64, 128, 12
    x = T.from_numpy(rng.normal(size=(B, N, C)).astype(np.float32)).cuda()
    We = T.from_numpy((rng.normal(size=(Co, 2 * C, 1, 1)) / 11).astype(np.float32)).cuda()
    ge, be = (T.from_numpy(rng.normal(size=Co).astype(np.float32)).cuda() for _ in range(2))
    idx = T.from_numpy(np.stack([np.stack([rng.permutation(N)[:k] for _ in range(N)]) for _ in range(B)]).astype(np.int32)).cuda()
    cote = T.from_numpy(rng.normal(size=(B, N, Co)).astype(np.float32)).cuda()

    def run_e(bf16):
        ops.TRAIN_BF16 = bf16
        try:
            xt, wt = x.clone().requires_grad_(True), We.clone().requires_grad_(True)
            out = hag.EdgeConvGN.apply(xt, idx, wt, ge, be, C, 2, 1e-5, 0.2)
            (out * cote).sum().backward()
            return out.detach().cpu().numpy(), xt.grad.cpu().numpy(), wt.grad.cpu().numpy()
        finally:
            ops.TRAIN_BF16 = False
    eb, ef = run_e(True), run_e(False)
    frac = np.mean(np.abs(eb[0] - ef[0]) <= 3e-2 * np.abs(ef[0]).max())
    assert frac > 0.995, frac                                  # a max over k may pick another neighbour after rounding
    _mostly_close(eb[2], ef[2], 5e-2, 0.995)               # EdgeConv dW bf16 vs fp32 (same allowance for switched slots)


def test_bf16_training_tracks_fp32(T):
    """BASELINE configs[4] in small: 50 AdamW steps on one synthetic batch in fp32 and with bf16 products, same
    initial weights and the same triplet draws: the two loss curves stay together and both go down."""
    from sednet_hip import ops, synth
    from sednet_hip.train import train_step
    from train_case import train_case
    x, labels, types, edges, edges_w, _ = train_case(synth, 2048, 2, seed0=500)
    batch = tuple(T.from_numpy(a).cuda() for a in (x, labels, types, edges, edges_w))
    curves = {}
    for bf16 in (False, True):
        ops.TRAIN_BF16 = bf16
        try:
            m = _model(T, 20, 3)
            opt = T.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0)
            hist = []
            for it in range(50):
                np.random.seed(5)
                hist.append(train_step(m, opt, batch, smoothing=0.025)["loss"])
            curves[bf16] = np.array(hist)
        finally:
            ops.TRAIN_BF16 = False
    f, b = curves[False], curves[True]
    assert np.isfinite(b).all() and b[-1] < 0.9 * b[0] and f[-1] < 0.9 * f[0], (f, b)
    assert abs(b[0] - f[0]) < 2e-2 * abs(f[0])                               # same weights: bf16 rounding only
    assert np.abs(b - f).max() < 0.15 * np.abs(f).max(), (f, b)             # the curves stay together
