"""SEDNet / DGCNNEncoderGn -- same constructor, state-dict keys and forward contract as
/root/reference/src/SEDNet.py:19-98 and :216-342, executed on gfx950 by libsedhip.so.

The torch.nn layers below are parameter containers only (so that reference checkpoints load unchanged,
SURVEY.md section 5); forward() never calls them. Activations live point-major [B,N,C] on the device:
    kNN (pairwise rows + exact selection) -> fused EdgeConv (gather + conv + GroupNorm statistics + max_k)
    x3 -> mlp1 with fused GroupNorm statistics + max over N -> head GEMMs with fused statistics.
With gradients enabled (training, train_sed_net.py:233-285) forward() runs the same kernels through the autograd
wrappers of sednet_hip/autograd.py (fused forward + HIP backward); under torch.no_grad() it runs the inference path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from sednet_hip import autograd as hag
from sednet_hip import ops


def _wt(conv, k_pad=None, cols=None):
    """Conv1d/Conv2d kernel-1 weight [Cout,Cin,...] -> transposed, zero padded [Kp, Coutp] + padded bias."""
    W = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
    if cols is not None:
        W = W[:, cols[0]:cols[1]]
    Cout, K = W.shape
    Kp = (K + 31) // 32 * 32 if k_pad is None else k_pad
    Coutp = (Cout + 63) // 64 * 64
    Wt = torch.zeros((Kp, Coutp), dtype=torch.float32, device=W.device)
    Wt[:K, :Cout] = W.t()
    b = None
    if conv.bias is not None:
        b = torch.zeros((Coutp,), dtype=torch.float32, device=W.device)
        b[:Cout] = conv.bias.detach().float()
    return Wt.contiguous(), b


class _PositionalEncoding1D(nn.Module):
    """State-dict stand-in for positional_encodings.PositionalEncoding1D(256), which the reference instantiates
    (SEDNet.py:285) and never calls: its persistent `inv_freq` buffer is part of real checkpoints."""

    def __init__(self, channels):
        super().__init__()
        channels = int(-(-channels // 2) * 2)
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels)))


class DGCNNEncoderGn(nn.Module):
    def __init__(self, mode=0, input_channels=3, nn_nb=80, normal_metric_W=1.):
        super(DGCNNEncoderGn, self).__init__()
        self.k = nn_nb
        self.dilation_factor = 1
        self.mode = mode
        self.drop = 0.0
        self.input_channels = input_channels
        self.normal_metric_W = normal_metric_W
        if self.mode == 0 or self.mode == 5:
            self.bn1 = nn.GroupNorm(2, 64)
            self.bn2 = nn.GroupNorm(2, 64)
            self.bn3 = nn.GroupNorm(2, 128)
            self.bn4 = nn.GroupNorm(4, 256)      # dead weights in the reference too (state-dict parity)
            self.bn5 = nn.GroupNorm(8, 1024)
            self.conv1 = nn.Sequential(nn.Conv2d(input_channels * 2, 64, kernel_size=1, bias=False), self.bn1,
                                       nn.LeakyReLU(negative_slope=0.2))
            self.conv2 = nn.Sequential(nn.Conv2d(64 * 2, 64, kernel_size=1, bias=False), self.bn2,
                                       nn.LeakyReLU(negative_slope=0.2))
            self.conv3 = nn.Sequential(nn.Conv2d(64 * 2, 128, kernel_size=1, bias=False), self.bn3,
                                       nn.LeakyReLU(negative_slope=0.2))
            self.mlp1 = nn.Conv1d(256, 1024, 1)
            self.bnmlp1 = nn.GroupNorm(8, 1024)
        self._cache = None
        self.graphs_in, self.keep_graphs = None, False      # test hooks of forward_point_major

    # -- weights in kernel layout --------------------------------------------------------------------
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _prepared(self):
        if self._cache is not None and self._cache.get("_sig") != self._signature():
            self._cache = None                      # parameters were replaced or modified in place
        if self._cache is None:
            c = {"_sig": self._signature()}
            for i, (conv, bn) in enumerate(((self.conv1[0], self.bn1), (self.conv2[0], self.bn2),
                                            (self.conv3[0], self.bn3)), 1):
                W = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
                C = W.shape[1] // 2
                g = bn.weight.detach().float().contiguous()
                c[f"e{i}"] = (W[:, :C].t().contiguous(), W[:, C:].t().contiguous(),
                              torch.where(g >= 0, 1.0, -1.0).float().contiguous(), g,
                              bn.bias.detach().float().contiguous(), bn.num_groups, bn.eps)
            c["mlp1"] = _wt(self.mlp1)
            c["bnmlp1"] = (self.bnmlp1.weight.detach().float().contiguous(),
                           self.bnmlp1.bias.detach().float().contiguous())
            self._cache = c
        return self._cache

    def _edge(self, key, x, C, idx, out_views, rowmax=None):
        """rowmax: row bounds raised by the LAST view's rows (the slice of `feats`)"""
        W1t, W2t, sgn, gamma, beta, G, eps = self._prepared()[key]
        ysel, stats = ops.edgeconv(x, C, idx, W1t, W2t, sgn, G, eps)
        for i, o in enumerate(out_views):
            ops.gn_apply(ysel, ysel.shape[2], G, stats, gamma, beta, ops.ACT_LEAKY, o, slope=0.2,
                         rowmax=rowmax if i == len(out_views) - 1 else None)

    def input_graph(self, x):
        """First-layer kNN graph: depends only on the input cloud (and k, W), not on the weights, so models that
        share k and normal_metric_W -- the type and instance models of the driver -- can share it."""
        return ops.knn_points_normals(x.detach().float().contiguous(), self.k, self.normal_metric_W)

    def input_order(self, x):
        """Morton order of the input cloud for the feature-space kNN sweeps (ops.spatial_order): like the input graph it depends on
        the cloud alone, so the two models of the driver share it. Speed only: the graphs do not depend on it."""
        return ops.spatial_order(x.detach().float().contiguous())

    def forward_point_major(self, x, idx1=None, feats_bound=None, order=None):
        """x [B,6,N] -> (x4 [B,1024], feats [B,N,256] point-major). idx1: optional precomputed input_graph(x).
        feats_bound: optional ops.row_bounds(B, N) that receives max |feats| per row (for the split-fp16 layers on feats).
        order: optional precomputed input_order(x)."""
        if self.mode != 5 or self.input_channels != 6:
            raise NotImplementedError("the HIP path implements mode 5 with xyz+normal input (the SED-Net configuration)")
        B, _, N = x.shape
        k = self.k
        x = x.detach().float().contiguous()
        dev = x.device
        feats = torch.empty((B, N, 256), dtype=torch.float32, device=dev)
        x8 = torch.zeros((B, N, 8), dtype=torch.float32, device=dev)
        x8[:, :, :6] = x.transpose(1, 2)
        # test hooks: `graphs_in` = three int32 [B,N,k] graphs used INSTEAD of the device's own (the reference's
        # torch.topk graphs -> the embedding then differs from the reference's only by arithmetic, not by k-th / (k+1)-th neighbour
        # ties); `keep_graphs` = True records the graphs of this forward in `last_graphs`
        gin = self.graphs_in
        idx = (ops.knn_points_normals(x, k, self.normal_metric_W) if idx1 is None else idx1) if gin is None else gin[0]
        if order is None and gin is None:
            order = ops.spatial_order(x)
        x1 = torch.empty((B, N, 64), dtype=torch.float32, device=dev)
        fb = feats_bound if feats_bound is not None else ops.row_bounds(B, N, dev)
        self._edge("e1", x8, 6, idx, (x1, feats[:, :, 0:64]), rowmax=fb)
        idx2 = ops.knn_features(x1, k, 64, order=order) if gin is None else gin[1]
        x2 = torch.empty((B, N, 64), dtype=torch.float32, device=dev)
        self._edge("e2", x1, 64, idx2, (x2, feats[:, :, 64:128]), rowmax=fb)
        idx3 = ops.knn_features(x2, k, 64, order=order) if gin is None else gin[2]
        self._edge("e3", x2, 64, idx3, (feats[:, :, 128:256],), rowmax=fb)
        if self.keep_graphs:
            self.last_graphs = (idx, idx2, idx3)
        Wt, b = self._prepared()["mlp1"]
        gamma, beta = self._prepared()["bnmlp1"]
        _, stats, colext = ops.pointwise(feats, Wt, 1024, bias=b, flags=ops.F_STATS | ops.F_COLEXT, G=8,
                                         eps=self.bnmlp1.eps, rowmax=fb)
        x4 = ops.colext_finalize(colext, B, N, 1024, 8, stats, gamma, beta)
        return x4, feats

    def forward_point_major_train(self, x, idx1=None):
        """Differentiable twin of forward_point_major (same kernels + recorded max-over-k slots)."""
        if self.mode != 5 or self.input_channels != 6:
            raise NotImplementedError("the HIP path implements mode 5 with xyz+normal input (the SED-Net configuration)")
        B, _, N = x.shape
        k = self.k
        x = x.detach().float().contiguous()
        x8 = torch.zeros((B, N, 8), dtype=torch.float32, device=x.device)
        x8[:, :, :6] = x.transpose(1, 2)
        idx = ops.knn_points_normals(x, k, self.normal_metric_W) if idx1 is None else idx1
        order = ops.spatial_order(x)                   # round 6: the ordered sweeps (speed only: the graphs do not depend on it)
        x1 = hag.edgeconv_gn(x8, idx, self.conv1[0], self.bn1, 6)
        idx2 = ops.knn_features(x1.detach(), k, 64, order=order)
        x2 = hag.edgeconv_gn(x1, idx2, self.conv2[0], self.bn2, 64)
        idx3 = ops.knn_features(x2.detach(), k, 64, order=order)
        x3 = hag.edgeconv_gn(x2, idx3, self.conv3[0], self.bn3, 64)
        self.last_graphs = (idx, idx2, idx3)          # the neighbour sets this step differentiated through
        feats = torch.cat([x1, x2, x3], dim=2)
        y = hag.conv_gn_act(feats, self.mlp1, self.bnmlp1)
        return y.max(dim=1)[0], feats

    def forward(self, x):
        """SEDNet.py:78-98 -> (x4 [B,1024], x_features [B,256,N])."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            x4, feats = self.forward_point_major_train(x)
        else:
            x4, feats = self.forward_point_major(x)
        return x4, feats.transpose(1, 2).contiguous()

    def load_state_dict(self, *a, **k):
        self._cache = None
        return super().load_state_dict(*a, **k)


class SEDNet(nn.Module):
    def __init__(self, emb_size=50, num_primitives=8, primitives=False, embedding=False, mode=0, num_channels=3,
                 loss_function=None, nn_nb=80, combine_label_prim=False, edge_module=False, late_fusion=False,
                 w_pos_enc=0.2, normal_metric_W=1., predict_normal=False):
        super(SEDNet, self).__init__()
        self.mode = mode
        self.encoder = DGCNNEncoderGn(mode=mode, input_channels=num_channels, nn_nb=nn_nb,
                                      normal_metric_W=normal_metric_W)
        self.drop = 0.0
        self.loss_function = loss_function
        self.w_pos_enc = w_pos_enc
        self.conv1 = torch.nn.Conv1d(1024 + 256, 512, 1)
        self.bn1 = nn.GroupNorm(8, 512)
        self.conv2 = torch.nn.Conv1d(512, 256, 1)
        self.bn2 = nn.GroupNorm(4, 256)
        self.emb_size = emb_size
        self.primitives = primitives
        self.embedding = embedding
        self.combine_label_prim = combine_label_prim
        self.late_fusion = late_fusion
        self.edge_module = None
        if edge_module:
            self.edge_module = nn.Sequential(torch.nn.Conv1d(256, 128, 1), nn.GroupNorm(4, 128),
                                             torch.nn.Conv1d(128, 2, 1))
        if self.combine_label_prim:
            self.asis = nn.Sequential(torch.nn.Conv1d(256, 256, 1), nn.GroupNorm(4, 256), nn.ReLU(True),
                                      nn.Dropout(0.0))
        if self.embedding:
            self.mlp_seg_prob1 = torch.nn.Conv1d(256, 256, 1)
            self.mlp_seg_prob2 = torch.nn.Conv1d(256, self.emb_size, 1)
            self.bn_seg_prob1 = nn.GroupNorm(4, 256)
        if primitives:
            self.mlp_prim_prob1 = torch.nn.Conv1d(256, 256, 1)
            self.mlp_prim_prob2 = torch.nn.Conv1d(256, num_primitives, 1)
            self.bn_prim_prob1 = nn.GroupNorm(4, 256)
        self.num_primitives = num_primitives
        self.predict_normal = predict_normal
        if predict_normal:
            raise NotImplementedError("predict_normal is off in every SED-Net script; not on the HIP path")
        self.pos_enc = _PositionalEncoding1D(256)
        self.prim_encoding = nn.Sequential(nn.Conv1d(8, 256, 1), nn.ReLU())
        self._cache = None

    def load_state_dict(self, state_dict, *a, **k):
        """Checkpoints written with or without the (unused) positional-encoding buffers both load strictly."""
        self._cache = None
        self.encoder._cache = None
        sd = {n: v for n, v in state_dict.items() if not n.startswith("pos_enc.") or n == "pos_enc.inv_freq"}
        sd.setdefault("pos_enc.inv_freq", self.pos_enc.inv_freq)
        return super().load_state_dict(sd, *a, **k)

    def _apply(self, fn, *a, **k):        # .cuda() / .to(): drop device-side weight caches
        self._cache = None
        self.encoder._cache = None
        return super()._apply(fn, *a, **k)

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _prepared(self):
        if self._cache is not None and self._cache.get("_sig") != self._signature():
            self._cache = None                      # parameters were replaced or modified in place
        if self._cache is None:
            gb = lambda m: (m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous())
            c = {
                "_sig": self._signature(),
                "conv1_g": self.conv1.weight.detach().float().reshape(512, 1280).contiguous(),
                "conv1_b": self.conv1.bias.detach().float().contiguous(),
                "conv1_f": _wt(self.conv1, cols=(1024, 1280))[0],
                "bn1": gb(self.bn1),
                "conv2": _wt(self.conv2), "bn2": gb(self.bn2),
                "prim1": _wt(self.mlp_prim_prob1), "bn_prim1": gb(self.bn_prim_prob1),
                "prim2": _wt(self.mlp_prim_prob2),
                "edge0": _wt(self.edge_module[0]), "edge1": gb(self.edge_module[1]), "edge2": _wt(self.edge_module[2]),
                "seg1": _wt(self.mlp_seg_prob1), "bn_seg1": gb(self.bn_seg_prob1), "seg2": _wt(self.mlp_seg_prob2),
                "asis0": _wt(self.asis[0]), "asis1": gb(self.asis[1]),
                "penc": _wt(self.prim_encoding[0]),
            }
            self._cache = c
        return self._cache

    def _conv_gn_relu(self, X, conv_key, bn_key, G, C, eps, act=ops.ACT_RELU, scale=1.0, addend=None, cbias=None,
                      Wt=None, bias=None, x_bound=None, y_bound=None, gn_in=None, lazy=False):
        """x_bound: the row bounds of X (-> the split-fp16 product); y_bound: row bounds to raise with the result's rows.
        gn_in: X is the pre-normalisation output of the layer in front, normalised by the GEMM while it loads (ops.pointwise).
        lazy: do not apply this layer's GroupNorm + activation -- return (Y raw, gn_in tuple for the consumers) instead of the
        normalised tensor (round 6: bn1 -> conv2 and bn2 -> prim1 / seg1 never materialise their outputs; same bits)."""
        c = self._prepared()
        if Wt is None:
            Wt, bias = c[conv_key]
        Y, stats, _ = ops.pointwise(X, Wt, C, bias=bias, cbias=cbias, flags=ops.F_STORE | ops.F_STATS, G=G, eps=eps,
                                    rowmax=x_bound, gn_in=gn_in)
        gamma, beta = c[bn_key]
        if lazy:
            return Y, (stats, gamma, beta, G, act)
        return ops.gn_apply(Y, C, G, stats, gamma, beta, act, Y, scale=scale, addend=addend, rowmax=y_bound)

    def forward_point_major(self, points, idx1=None, order=None):
        """points [B,6,N] -> (embedding [B,N,emb], log_prob [B,N,P], edges [B,N,2]) point-major device tensors
        (views into kernel output buffers). This is what the batched driver consumes: no transposes.
        idx1: optional first-layer graph from encoder.input_graph(points) (shared between models);
        order: optional encoder.input_order(points) (likewise; speed only)."""
        if not (self.primitives and self.embedding and self.edge_module is not None and self.combine_label_prim
                and self.late_fusion):
            raise NotImplementedError("the HIP path implements the configuration used by the SED-Net scripts "
                                      "(embedding, primitives, edge_module, combine_label_prim, late_fusion)")
        with torch.no_grad():
            c = self._prepared()
            B, _, N = points.shape
            dev = points.device
            # per-row magnitude bounds of the GroupNorm outputs: what lets the next layer run in the two-plane split-fp16 form
            bnd = [ops.row_bounds(B, N, dev) for _ in range(5)]
            x4, feats = self.encoder.forward_point_major(points, idx1, feats_bound=bnd[0], order=order)
            # conv1 over cat(repeat(x4), feats): the repeated-global part is a per-cloud bias   (:300-303)
            cb = ops.gemv_bias(c["conv1_g"], 1280, 1024, c["conv1_b"], x4)
            # bn1, bn2, bn_prim_prob1 and the edge module's norm have GEMM consumers only: those apply them while loading (round 6)
            fold = ops.gn_in_ok(512, 256) and ops.gn_in_ok(256, 64)
            if fold:
                y1, g1 = self._conv_gn_relu(feats, None, "bn1", 8, 512, self.bn1.eps, cbias=cb, Wt=c["conv1_f"], lazy=True)
                x_all, g_all = self._conv_gn_relu(y1, "conv2", "bn2", 4, 256, self.bn2.eps, gn_in=g1, lazy=True)       # :304
            else:
                a1 = self._conv_gn_relu(feats, None, "bn1", 8, 512, self.bn1.eps, cbias=cb, Wt=c["conv1_f"],
                                        x_bound=bnd[0], y_bound=bnd[1])
                x_all = self._conv_gn_relu(a1, "conv2", "bn2", 4, 256, self.bn2.eps, x_bound=bnd[1], y_bound=bnd[2])   # :304
                g_all = None
            # (bn_prim_prob1's output feeds mlp_prim_prob2, the edge module and asis; the edge module's norm its last conv: GEMMs only)
            layer = lambda *a, **k: self._conv_gn_relu(*a, lazy=True, **k) if fold else (self._conv_gn_relu(*a, **k), None)
            x_type, g_type = layer(x_all, "prim1", "bn_prim1", 4, 256, self.bn_prim_prob1.eps,
                                   x_bound=bnd[2], y_bound=bnd[3], gn_in=g_all)                       # :311
            P = self.num_primitives
            te = torch.zeros((B, N, 32), dtype=torch.float32, device=dev)       # cat(type_logit, edges), K padded
            Wt, b = c["prim2"]
            ops.pointwise(x_type, Wt, P, bias=b, out=te[:, :, 0:P], gn_in=g_type)                    # :312
            log_prob = ops.log_softmax_rows(te, P)                                                      # :313
            e1, g_e1 = layer(x_type, "edge0", "edge1", 4, 128, self.edge_module[1].eps, act=ops.ACT_NONE,
                             x_bound=bnd[3], gn_in=g_type)
            Wt, b = c["edge2"]
            ops.pointwise(e1, Wt, 2, bias=b, out=te[:, :, P:P + 2], gn_in=g_e1)                      # :316-317
            Wt, b = c["penc"]
            if fold:
                # bn_seg_prob1, asis' norm + its residual add and the position-encoding add (:320-326) in ONE elementwise pass
                ys, gs = layer(x_all, "seg1", "bn_seg1", 4, 256, self.bn_seg_prob1.eps, gn_in=g_all)              # :320
                ya, ga = layer(x_type, "asis0", "asis1", 4, 256, self.asis[1].eps, gn_in=g_type)                     # :322
                pe, _, _ = ops.pointwise(te, Wt, 256, bias=b, flags=ops.F_STORE | ops.F_RELU)         # :326
                x = ops.gn_apply_fused(ya, ga, self.w_pos_enc, ys, gs, pe, self.w_pos_enc, pe)
            else:
                xs = self._conv_gn_relu(x_all, "seg1", "bn_seg1", 4, 256, self.bn_seg_prob1.eps, x_bound=bnd[2], gn_in=g_all)   # :320
                x = self._conv_gn_relu(x_type, "asis0", "asis1", 4, 256, self.asis[1].eps,
                                       scale=self.w_pos_enc, addend=xs, x_bound=bnd[3], gn_in=g_type)      # :322
                pe, _, _ = ops.pointwise(te, Wt, 256, bias=b, flags=ops.F_STORE | ops.F_RELU)             # :326
                x = ops.gn_apply(pe, 256, 0, None, None, None, ops.ACT_NONE, pe, scale=self.w_pos_enc, addend=x, rowmax=bnd[4])
            Wt, b = c["seg2"]
            emb, _, _ = ops.pointwise(x, Wt, self.emb_size, bias=b, rowmax=bnd[4])                    # :329
        return emb, log_prob, te[:, :, P:P + 2]

    def forward_point_major_train(self, points, idx1=None):
        """Differentiable twin of forward_point_major: the Conv+GroupNorm+activation layers run the fused HIP forward
        and the HIP backward (sednet_hip/autograd.py); the three thin output projections (256->6, 128->2, 8->256,
        256->emb) are plain library GEMMs."""
        if not (self.primitives and self.embedding and self.edge_module is not None and self.combine_label_prim
                and self.late_fusion):
            raise NotImplementedError("the HIP path implements the configuration used by the SED-Net scripts "
                                      "(embedding, primitives, edge_module, combine_label_prim, late_fusion)")
        x4, feats = self.encoder.forward_point_major_train(points, idx1)
        Wc = self.conv1.weight.reshape(512, 1280)
        cb = F.linear(x4, Wc[:, :1024], self.conv1.bias)                  # repeated-global part = per-cloud bias :300-303
        a1 = hag.conv_gn_act(feats, self.conv1, self.bn1, cbias=cb, weight=Wc[:, 1024:], bias=None)
        x_all = hag.conv_gn_act(a1, self.conv2, self.bn2)                                              # :304
        x_type = hag.conv_gn_act(x_all, self.mlp_prim_prob1, self.bn_prim_prob1)                      # :311
        lin = lambda t, conv: F.linear(t, conv.weight.reshape(conv.weight.shape[0], -1), conv.bias)
        type_logit = lin(x_type, self.mlp_prim_prob2)                                                  # :312
        log_prob = F.log_softmax(type_logit, dim=2)
        e1 = hag.conv_gn_act(x_type, self.edge_module[0], self.edge_module[1], act=ops.ACT_NONE)
        edges = lin(e1, self.edge_module[2])                                                           # :316-317
        xs = hag.conv_gn_act(x_all, self.mlp_seg_prob1, self.bn_seg_prob1)                            # :320
        x = self.w_pos_enc * hag.conv_gn_act(x_type, self.asis[0], self.asis[1]) + xs                  # :322
        pe = F.relu(lin(torch.cat([type_logit.detach(), edges.detach()], dim=2), self.prim_encoding[0]))
        x = x + self.w_pos_enc * pe                                                                    # :326
        return lin(x, self.mlp_seg_prob2), log_prob, edges                                             # :329

    def forward(self, points, labels=None, compute_loss=False):
        """SEDNet.py:292-342 -> [embedding [B,emb,N], log_prob [B,P,N], embed_loss [1], edges [B,2,N]]."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            emb, log_prob, edges = self.forward_point_major_train(points)
        else:
            emb, log_prob, edges = self.forward_point_major(points)
        embedding = emb.transpose(1, 2).contiguous()
        primitives_log_prob = log_prob.transpose(1, 2).contiguous()
        edges_pred = edges.transpose(1, 2).contiguous()
        if compute_loss:
            embed_loss = self.loss_function(embedding, labels.data.cpu().numpy())                       # :332-333
        else:
            embed_loss = torch.zeros(1, device=points.device)                                          # :335
        return [embedding, primitives_log_prob, embed_loss, edges_pred]
