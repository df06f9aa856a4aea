"""torch.autograd wrappers of the fused HIP layers -- the training step of SURVEY section 8 f-3.

The reference differentiates Conv -> GroupNorm -> activation [-> max over k] with torch.autograd over materialised
[B,2C,N,k] / [B,Cout,N,k] tensors (/root/reference/src/SEDNet.py:37-45,78-98,300-329; train_sed_net.py:272).
Here the forward is the fused HIP kernel of the inference path (plus the selected-slot record), the backward is
edgeconv_bwd.hip: GroupNorm's backward reduced to  dy = S [j == j*] + alpha_g + kappa_g y  and, for EdgeConv, two
kernels that recompute y tile by tile instead of storing it. The two GEMMs of a pointwise layer's backward
(dX = dy W, dW = dy^T X) run on the repo's own MFMA GEMM (gemm.hip; round 1 called rocBLAS through torch.matmul).
ops.TRAIN_BF16 = True switches the forward products of the 64-channel EdgeConv layers and of every pointwise layer, and
both backward GEMMs, to bf16 (operands rounded while staged, fp32 accumulate; weights, activations in memory, GroupNorm
statistics and the EdgeConv backward stay fp32) -- BASELINE configs[4].

Activations are point-major [B,N,C]; indices are not differentiated (topk indices carry no gradient in the reference
either).
"""
import torch

from . import ops


def _wt_pad(W2d):
    """[Cout,K] weight -> transposed, zero padded [Kp, Coutp] for ops.pointwise."""
    Cout, K = W2d.shape
    Kp = (K + 31) // 32 * 32
    Coutp = (Cout + 63) // 64 * 64
    Wt = torch.zeros((Kp, Coutp), dtype=torch.float32, device=W2d.device)
    Wt[:K, :Cout] = W2d.t()
    return Wt


class EdgeConvGN(torch.autograd.Function):
    """out[B,N,Cout] = max_k LeakyReLU(GN(Conv2d(cat(x_j - x_i, x_i)))); x [B,N,ldx] with C real channels."""

    @staticmethod
    def forward(ctx, x, idx, weight, gamma, beta, C, G, eps, slope):
        W = weight.detach().float().reshape(weight.shape[0], -1)
        W1t = W[:, :C].t().contiguous()
        W2t = W[:, C:].t().contiguous()
        g = gamma.detach().float().contiguous()
        b = beta.detach().float().contiguous()
        sgn = torch.where(g >= 0, 1.0, -1.0).float().contiguous()
        xd = x.detach().contiguous()
        ysel, stats, jsel = ops.edgeconv_train(xd, C, idx, W1t, W2t, sgn, G, eps, bf16=ops.TRAIN_BF16)
        out = torch.empty_like(ysel)
        ops.gn_apply(ysel, ysel.shape[2], G, stats, g, b, ops.ACT_LEAKY, out, slope=slope)
        ctx.save_for_backward(xd, idx, W1t, W2t, g, b, ysel, stats, jsel)
        ctx.meta = (C, G, slope, weight.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        xd, idx, W1t, W2t, g, b, ysel, stats, jsel = ctx.saved_tensors
        C, G, slope, wshape = ctx.meta
        B, N, Cout = ysel.shape
        k = idx.shape[2]
        S, dgamma, dbeta, ak = ops.gn_bwd_reduce(dout.contiguous(), ysel, Cout, G, float(Cout // G) * N * k, stats, g, b,
                                                 ops.ACT_LEAKY, slope)
        dW1t, dW2t, dx = ops.edgeconv_bwd(xd, C, idx, W1t, W2t, G, S, jsel, ak, ctx.needs_input_grad[0],
                                          bf16=ops.TRAIN_BF16)
        dW = torch.cat([dW1t.t(), dW2t.t()], dim=1).reshape(wshape)
        return dx, None, dW, dgamma, dbeta, None, None, None, None


class ConvGNAct(torch.autograd.Function):
    """out[B,N,Cout] = act(GN(X W^T + bias + cbias[b])); X [B,N,K] (row-strided view allowed), W [Cout,K(,1)]."""

    @staticmethod
    def forward(ctx, X, weight, bias, cbias, gamma, beta, G, eps, act):
        W = weight.detach().float().reshape(weight.shape[0], -1)
        Cout, K = W.shape
        Xd = X.detach()
        if Xd.stride(2) != 1 or Xd.stride(0) != Xd.shape[1] * Xd.stride(1):
            Xd = Xd.contiguous()
        if Xd.shape[2] % 32 != 0:
            raise ValueError("ConvGNAct: input width must be a multiple of 32")
        g = gamma.detach().float().contiguous()
        b = beta.detach().float().contiguous()
        bp = None
        if bias is not None:
            bp = torch.zeros(((Cout + 63) // 64 * 64,), dtype=torch.float32, device=W.device)
            bp[:Cout] = bias.detach().float()
        cb = cbias.detach().float().contiguous() if cbias is not None else None
        Y, stats, _ = ops.pointwise(Xd, _wt_pad(W), Cout, bias=bp, cbias=cb, flags=ops.F_STORE | ops.F_STATS, G=G,
                                    eps=eps, bf16=ops.TRAIN_BF16, split=False)      # weights change every step: no cached split
        out = torch.empty((Y.shape[0], Y.shape[1], Cout), dtype=torch.float32, device=Y.device)
        ops.gn_apply(Y, Cout, G, stats, g, b, act, out)
        ctx.save_for_backward(Xd, W, g, b, Y, stats)
        ctx.meta = (G, act, weight.shape, bias is not None, cbias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        Xd, W, g, b, Y, stats = ctx.saved_tensors
        G, act, wshape, has_bias, has_cbias = ctx.meta
        B, N, Cout = dout.shape
        S, dgamma, dbeta, ak = ops.gn_bwd_reduce(dout.contiguous(), Y, Cout, G, float(Cout // G) * N, stats, g, b, act)
        dy = ops.gn_bwd_apply(S, Y, Cout, G, ak)
        K = W.shape[1]
        dy2 = dy.reshape(B * N, Cout)
        dX = ops.gemm(dy2, W).reshape(B, N, K) if ctx.needs_input_grad[0] else None            # dX = dy W
        X2 = Xd.as_strided((B * N, K), (Xd.stride(1), 1), Xd.storage_offset())               # [P, K] view of the saved input
        dW = ops.gemm(dy2, X2, transA=True).reshape(wshape)                                  # dW = dy^T X
        dcb = dy.sum(1) if has_cbias else None
        dbias = (dcb.sum(0) if has_cbias else dy.sum((0, 1))) if has_bias else None
        return dX, dW, dbias, dcb, dgamma, dbeta, None, None, None


def edgeconv_gn(x, idx, conv, bn, C, slope=0.2):
    return EdgeConvGN.apply(x, idx, conv.weight, bn.weight, bn.bias, C, bn.num_groups, bn.eps, slope)


def conv_gn_act(X, conv, bn, act=ops.ACT_RELU, cbias=None, weight=None, bias="conv"):
    """conv: nn.Conv1d (kernel 1); weight / bias override (conv1's feature columns with the global part as cbias)."""
    W = conv.weight if weight is None else weight
    bs = conv.bias if isinstance(bias, str) else bias
    return ConvGNAct.apply(X, W, bs, cbias, bn.weight, bn.bias, bn.num_groups, bn.eps, act)
