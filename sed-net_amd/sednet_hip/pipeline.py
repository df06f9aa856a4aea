"""Batched, device-resident counterpart of the reference's per-cloud inference loop
(/root/reference/generate_predictions_aug.py:213-387) extended by the primitive-fit stage whose reference
caller is Fitting_patches_and_edges/residual_utils.py:86-331:

    type model  -> per-point primitive type (argmax of log-probs)               :224-226, :365
    inst model  -> per-point embedding -> unit rows                            :227-229, :380
    guard_mean_shift(quantile 0.015, 50 iterations, x1.2 while > 49 clusters)   :25-35, :382
    per segment: type vote (stats.mode) -> LSQ fit -> closed-form residual     residual_utils.py:259, :300-331

B clouds go through every stage in one launch each. Host syncs per step, all of them copies of a few bytes per cloud: the
overflow flags of the five streaming kNN calls, read together once after both forwards (a raised flag repeats the forwards
on the exact path), and per guard pass the bandwidth kernel's per-cloud overflow flags, the density probe that picks the mean-shift
schedule (ops.ms_near_fraction) and the cluster counts the guard loop branches on (the reference's :31; it syncs at least
three times per CLOUD and pass, plus a numpy round trip).
"""
import os

import torch

from . import ops


class _StageTimer:
    """records (stage, start event, end event) on the current stream when the pipeline's `stage_times` list is set"""

    def __init__(self, sink):
        self.sink = sink
        if sink is not None:
            self.last = torch.cuda.Event(enable_timing=True)
            self.last.record()

    def mark(self, name):
        if self.sink is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.sink.append((name, self.last, e))
            self.last = e


class SegmentationPipeline:
    stage_times = None            # set to a list to collect per-stage event pairs (bench.py)
    # up to this many clouds per call the two models' forwards run on two HIP streams (tools/small_batch_streams.py: 1.07 x at
    # one cloud per call, 1.10 x at 2, 1.06 x at 4, 1.02 x at 8, 1.01 x at 16; beyond, every kernel fills the chip on its own)
    TWO_STREAM_MAX_CLOUDS = 16
    _side = None
    _spec = None
    OVERLAP_SPECTRAL = True       # HPNet flow: the network-independent spectral block on its own stream beside the forwards
    _graphs = None
    _graph_ok = True

    def __init__(self, model_type, model_inst, quantile=0.015, iterations=50, max_segments=50, fit=True, dist=None, *,
                 hpnet):
        """dist: an initialised torch.distributed (world > 1) -> the guard loop's retry passes are balanced over the
        ranks (every rank must then call the pipeline the same number of times). hpnet: a REQUIRED keyword (rounds 1-3
        defaulted to False, round 4 to True -- the choice changes labels, embedding width and step time, so the caller states it;
        True = the reference's HPNet_embed = True and this repo's generate_predictions default, bench.py's headline leg passes False): the script's
        spectral re-weighting of the embedding between the instance model and the clustering (generate_predictions_aug.py:58,
        :371-377; src/smooth_normal_matrix.hpnet_process): the embedding grows to 140 columns (160 padded)."""
        from src.mean_shift import MeanShift
        self.model_type, self.model_inst = model_type, model_inst
        self.quantile, self.iterations, self.S, self.fit = quantile, iterations, max_segments, fit
        self.ms, self.dist, self.hpnet = MeanShift(), dist, hpnet

    def _forwards(self, x6, ev=None):
        """both models on x6 -> (log_prob, t_model, emb, edges, X, overflow flags)"""
        if (self.GRAPH_FORWARDS and ops.FUSED_KNN and x6.shape[0] <= self.TWO_STREAM_MAX_CLOUDS and ev is not None
                and ev.sink is None and self._graph_ok):
            try:
                return self._graph_forwards(x6)
            except Exception as e:                                   # capture not possible here: the two-stream order below
                import warnings
                warnings.warn(f"SegmentationPipeline: HIP-graph capture of the forwards failed ({e!r}); using the two-stream order")
                self._graph_ok = False
                self._graphs = None          # nothing half-captured is kept; the forked side stream is replaced by a fresh one
                self._side = None
        return self._forwards_plain(x6, ev)

    def _forwards_plain(self, x6, ev=None):
        flags = []
        prev, ops.DEFERRED_KNN_FLAGS = ops.DEFERRED_KNN_FLAGS, flags
        try:
            # the first-layer kNN graph depends only on the cloud: computed once for both models when they agree on (k, W)
            e0, e1 = self.model_type.encoder, self.model_inst.encoder
            idx1 = e0.input_graph(x6) if (e0.k == e1.k and e0.normal_metric_W == e1.normal_metric_W) else None
            order = e0.input_order(x6)                      # row order of the feature-space kNN sweeps: a function of the cloud alone
            if ev is not None:
                ev.mark("input_graph")
            if x6.shape[0] <= self.TWO_STREAM_MAX_CLOUDS:
                # Few clouds per call: a forward's kernels are 79-workgroup grids per cloud that leave most of the 256 CUs idle,
                # and the two models are independent until the type vote -- the type model runs on a side stream beside the
                # instance model (every op launches on torch's current stream). One cloud per call: 10.6 -> 9.9 ms (the host enqueues
                # the two forwards one after the other, so the overlap is partial; enqueuing the type model from a second host thread is
                # slower, 12.5 ms: the wrappers are Python-bound and fight over the GIL).
                main = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream()
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    _, log_prob, _ = self.model_type.forward_point_major(x6, idx1, order)
                    t_model = ops.row_argmax(log_prob, log_prob.shape[2])
                emb, _, edges = self.model_inst.forward_point_major(x6, idx1, order)
                X = ops.row_normalize(emb, emb.shape[2])
                main.wait_stream(self._side)
                log_prob.record_stream(main)
                t_model.record_stream(main)
                if ev is not None:
                    ev.mark("type_model")
            else:
                _, log_prob, _ = self.model_type.forward_point_major(x6, idx1, order)
                t_model = ops.row_argmax(log_prob, log_prob.shape[2])
                if ev is not None:
                    ev.mark("type_model")
                emb, _, edges = self.model_inst.forward_point_major(x6, idx1, order)
                X = ops.row_normalize(emb, emb.shape[2])
        finally:
            ops.DEFERRED_KNN_FLAGS = prev
        return log_prob, t_model, emb, edges, X, flags

    # ---- few clouds per call: both forwards as ONE HIP graph with the type model forked onto a side stream -----------------
    # Replaying removes the host from the picture: the two models' kernels (79-workgroup grids at one cloud) are queued on
    # their two streams at once and really run side by side -- the plain two-stream order is limited by the host enqueuing one
    # forward after the other. One graph per input shape; the outputs live in the graph's memory pool and are copied out.
    GRAPH_FORWARDS = os.environ.get("SED_GRAPH_FORWARDS", "1") != "0"      # 0: debugging runs without the caching allocator (no capture possible there)

    def _graph_forwards(self, x6):
        # (a graph bakes in the addresses of the models' cached weight images: keyed by the parameters' identity / version, and the
        # entry keeps those caches alive)
        # (SEDNet._signature covers the encoder's parameters too: it walks self.parameters())
        # ... and every kernel-selection switch of the forwards that the capture bakes in: toggling one replays nothing stale
        key = (tuple(x6.shape), self.model_type._signature(), self.model_inst._signature(),
               ops.POINTWISE_SPLIT, ops.EDGECONV_SPLIT, ops.FUSED_KNN, ops.TRAIN_BF16)
        ent = self._graphs.get(key) if self._graphs is not None else None
        if ent is None:
            if self._graphs is None:
                self._graphs = {}
            if self._side is None:
                self._side = torch.cuda.Stream()
            static_x = x6.clone()
            self._forwards_plain(static_x)                        # warm-up outside the capture: weight images, kernel attributes
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            flags = []
            prev, ops.DEFERRED_KNN_FLAGS = ops.DEFERRED_KNN_FLAGS, flags
            try:
                with torch.cuda.graph(g):
                    cap = torch.cuda.current_stream()
                    e0, e1 = self.model_type.encoder, self.model_inst.encoder
                    idx1 = e0.input_graph(static_x) if (e0.k == e1.k and e0.normal_metric_W == e1.normal_metric_W) else None
                    order = e0.input_order(static_x)
                    self._side.wait_stream(cap)
                    with torch.cuda.stream(self._side):
                        _, log_prob, _ = self.model_type.forward_point_major(static_x, idx1, order)
                        t_model = ops.row_argmax(log_prob, log_prob.shape[2])
                    emb, _, edges = self.model_inst.forward_point_major(static_x, idx1, order)
                    X = ops.row_normalize(emb, emb.shape[2])
                    cap.wait_stream(self._side)
            finally:
                ops.DEFERRED_KNN_FLAGS = prev
            keep = (self.model_type._prepared(), self.model_type.encoder._prepared(), self.model_inst._prepared(),
                    self.model_inst.encoder._prepared())
            if len(self._graphs) >= 4:                            # a handful of shapes at most; older graphs are dropped
                self._graphs.pop(next(iter(self._graphs)))
            ent = (g, static_x, (log_prob, t_model, emb, edges, X), flags, keep)
            self._graphs[key] = ent
        g, static_x, outs, flags = ent[:4]
        static_x.copy_(x6)
        g.replay()
        log_prob, t_model, emb, edges, X = (o.clone() for o in outs)
        return log_prob, t_model, emb, edges, X, flags

    def _forwards_checked(self, x6, ev=None):
        """the streaming kNN kernels' overflow flags are read ONCE for the whole step; a raised flag (masses of duplicate
        points) repeats the forwards on the exact materialised path. (Replaying the forwards from a HIP graph was tried for
        the one-cloud-per-call pattern and changes nothing -- 5.85 vs 5.83 ms: at B = 1 they are bound by the run time of
        single workgroups sweeping all key tiles, not by launches.)"""
        log_prob, t_model, emb, edges, X, flags = self._forwards(x6, ev)
        if not ops.deferred_knn_overflow(flags):
            ops.FUSED_STATS["fused"] += len(flags)
            return log_prob, t_model, emb, edges, X
        ops.FUSED_STATS["fallback"] += len(flags)
        prev, ops.FUSED_KNN = ops.FUSED_KNN, False
        try:
            return self._forwards(x6, ev)[:5]
        finally:
            ops.FUSED_KNN = prev

    @torch.no_grad()
    def __call__(self, x6, embedding=None, types=None):
        """x6 [B,6,N] (xyz + unit normals, channel-major like SEDNet.forward) -> dict of device tensors.
        embedding [B,N,d] / types [B,N] i32: replace the instance model's embedding / the type model's argmax AFTER both
        forwards have run (bench.py's realistic leg and the full-size tests plant cluster structure that closed-form
        weights cannot produce; with real checkpoints leave them None)."""
        ev = _StageTimer(self.stage_times)
        x6 = x6.float().contiguous()
        spectral = None
        if self.hpnet and self.OVERLAP_SPECTRAL:
            # The spectral block of the HPNet stage (far-kNN graph, affinity CSR, 10 LOBPCG iterations, eigenvector entropy: ~45 of
            # the stage's 60 ms per 64 clouds) depends on xyz and normals only -- not on either network. It is enqueued FIRST, on its
            # own stream, and runs beside the two forwards: its kernels are latency-bound (one-wave Ritz problems, tall-skinny Gram
            # products, CSR gathers) and leave the CUs to the backbone. Nothing inside synchronises with the host.
            from src.smooth_normal_matrix import hpnet_spectral
            main = torch.cuda.current_stream()
            if self._spec is None:
                # high priority: the chain is ~400 small launches whose workgroups should slip in between the forwards' large grids
                self._spec = torch.cuda.Stream(priority=-1)
            pts_h, nrm_h = x6[:, 0:3].transpose(1, 2).contiguous(), x6[:, 3:6].transpose(1, 2).contiguous()
            self._spec.wait_stream(main)
            far_flags = []
            with torch.cuda.stream(self._spec):
                #  the chain's own duration, taken on ITS stream: it overlaps the forwards, so it appears in stage_times as an
                # extra entry ("hpnet_spectral_overlapped") beside the main stream's stages, whose sum stays the step; the "hpnet" stage
                # below is only what is left after the join (feature entropy + concatenation). Memory: the chain's intermediates
                # (far-kNN workspaces, CSR, LOBPCG blocks for all clouds of the call) live in this stream's own pool of the caching
                # allocator and are not reused by the main stream -- ~1.2 GB at 64 x 10 000 points on top of the step's ~9 GB.
                if self.stage_times is not None:
                    s0 = torch.cuda.Event(enable_timing=True)
                    s0.record()
                spectral = hpnet_spectral(pts_h, nrm_h, 0.5, 1000, flags=far_flags)
                if self.stage_times is not None:
                    s1 = torch.cuda.Event(enable_timing=True)
                    s1.record()
                    self.stage_times.append(("hpnet_spectral_overlapped", s0, s1))
        log_prob, t_model, emb, edges, X = self._forwards_checked(x6, ev)
        ops.finite_canary("forwards", {"log_prob": log_prob, "embedding": emb, "edges": edges}, {"x6": x6})
        ops.finite_canary("row_normalize", {"X": X}, {"embedding": emb})
        if embedding is not None:
            X = ops.row_normalize(embedding.float().contiguous(), embedding.shape[2])
        types = t_model if types is None else types.int().contiguous()
        ev.mark("instance_model")
        if self.hpnet:
            from src.smooth_normal_matrix import hpnet_process
            if spectral is not None:
                main.wait_stream(self._spec)
                for t in spectral:
                    t.record_stream(main)
                ops.knn_farthest_check(far_flags)
            wide = hpnet_process(emb if embedding is None else embedding.float(), x6[:, 0:3].transpose(1, 2).contiguous(),
                                 x6[:, 3:6].transpose(1, 2).contiguous(), normal_smooth_w=0.5, CHUNK=1000,
                                 spectral=spectral)                                                           # :59, :375
            X = ops.row_normalize(wide.contiguous(), wide.shape[2])                                           # :377
            ops.finite_canary("hpnet", {"wide": wide, "X": X}, {"embedding": emb, "x6": x6})
            ev.mark("hpnet")
        labels, bw, n_labels, passes = self.ms.guard_mean_shift_batch(X, self.quantile, self.iterations, dist=self.dist)
        ev.mark("mean_shift")
        out = {"labels": labels, "types": types, "bw": bw, "n_labels": n_labels, "passes": passes, "edges": edges}
        if self.fit:
            seg_type, seg_count = ops.segment_type_vote(labels, types, self.S, log_prob.shape[2])
            pts = x6[:, 0:3].transpose(1, 2).contiguous()
            nrm = x6[:, 3:6].transpose(1, 2).contiguous()
            params, valid = ops.fit_segments(pts, nrm, seg_type, labels=labels)
            _, seg_res = ops.residual_segments(pts, seg_type, params, valid, labels=labels, sqrt=True, per_point=False)
            out.update(seg_type=seg_type, seg_count=seg_count, params=params, valid=valid, seg_residual=seg_res)
            ops.finite_canary("fits", {"params": torch.where(valid.bool().unsqueeze(-1), params, torch.zeros_like(params)),
                                       "seg_residual": torch.where(valid.bool(), seg_res, torch.zeros_like(seg_res))},
                              {"x6": x6, "labels": labels, "types": types})
            ev.mark("fits")
        return out
