"""MeanShift -- same operator surface as /root/reference/src/mean_shift.py:11-179, executed by the
gfx950 kernels in libsedhip.so (no N x N tensor is materialised by the iterations or the NMS, and the
NMS never leaves the device).

Single-cloud methods keep the reference signatures; `*_batch` variants take [B,N,d] and are what the
batched driver uses. Only the gaussian kernel is implemented (the `epa` branch at :64-68 is dead code in
the reference's callers).
"""
import numpy as np
import torch

from sednet_hip import ops


def _inference_only(t, what):
    """The reference's mean_shift is written with differentiable torch ops (and EmbeddingLoss(if_mean_shift=True) relies
    on it); the HIP kernels have no backward. Fail loudly instead of returning a tensor without grad_fn."""
    if torch.is_grad_enabled() and torch.is_tensor(t) and t.requires_grad:
        raise NotImplementedError(f"{what} runs on the HIP inference kernels and is not differentiable: call it under "
                                  "torch.no_grad() or on a detached tensor")


def _as_bw_tensor(b, B, device):
    if torch.is_tensor(b):
        return b.to(device=device, dtype=torch.float32).reshape(-1).expand(B).contiguous()
    return torch.full((B,), float(b), dtype=torch.float32, device=device)


class MeanShift:
    match_rng = False          # batched calls: replay the reference's per-cloud np.random.shuffle even when it cannot matter

    def __init__(self):
        pass

    # ------------------------------------------------------------------ reference surface (one cloud)
    def mean_shift(self, X, num_samples, quantile, iterations, kernel_type="gaussian", bw=None, nms=True):
        """mean_shift.py:19-43 -> (new_X, center, bw, labels) or (new_X, bw) when nms=False."""
        self._check_kernel(kernel_type)
        _inference_only(X, "MeanShift.mean_shift")
        Xp = ops.pad_features(X.detach())[None]
        d = X.shape[1]
        if bw is None:
            bw = self._bandwidth_padded(Xp[0], num_samples, quantile)       # includes clamp(min=0.003), :34
        bwt = _as_bw_tensor(bw, 1, X.device)
        new_Xp = ops.ms_iterate(Xp, bwt, iterations)
        new_X = new_Xp[0, :, :d]
        bw_out = bwt[0] if not torch.is_tensor(bw) or bw.dim() == 0 else bw
        if not nms:
            return new_X, bw_out
        labels, ids, n_c, _ = ops.ms_nms(new_Xp, Xp, bwt)
        m = int(n_c[0].item())
        indices = ids[0, :m].long()
        return new_X, new_X[indices], bw_out, labels[0].long()

    def mean_shift_(self, X, b, iterations=10, kernel_type="gaussian"):
        """mean_shift.py:45-79 -> (new_X, X)."""
        self._check_kernel(kernel_type)
        _inference_only(X, "MeanShift.mean_shift_")
        Xp = ops.pad_features(X.detach())[None]
        new_X = ops.ms_iterate(Xp, _as_bw_tensor(b, 1, X.device), iterations)[0, :, :X.shape[1]]
        return new_X, X

    def guard_mean_shift(self, embedding, quantile, iterations, kernel_type="gaussian"):
        """mean_shift.py:81-96 (class variant: 5000 samples, quantile doubles)."""
        while True:
            _, center, bandwidth, cluster_ids = self.mean_shift(
                embedding, 5000, quantile, iterations, kernel_type=kernel_type)
            if torch.unique(cluster_ids).shape[0] > 49:
                quantile *= 2
            else:
                break
        return center, bandwidth, cluster_ids

    def compute_bandwidth(self, X, num_samples, quantile):
        """mean_shift.py:115-137 (without the caller's clamp) -> 0-dim tensor."""
        return self._bandwidth_padded(ops.pad_features(X.detach()), num_samples, quantile, min_bw=0.0)

    def nms(self, centers, X, b):
        """mean_shift.py:139-179 -> (centers[ids], ids, labels)."""
        Cp, Xp = ops.pad_features(centers.detach())[None], ops.pad_features(X.detach())[None]
        labels, ids, n_c, _ = ops.ms_nms(Cp, Xp, _as_bw_tensor(b, 1, X.device))
        indices = ids[0, :int(n_c[0].item())].long()
        return centers[indices], indices, labels[0].long()

    # ------------------------------------------------------------------ batched (B clouds per call)
    def mean_shift_batch(self, X, num_samples, quantile, iterations, bw=None):
        """X [B,N,d] unit rows -> (new_X [B,N,d], bw [B], labels [B,N] i32, centre_ids [B,N] i32,
        n_centres [B] i32, n_labels [B] i32); nothing is copied to the host."""
        B, N, d = X.shape
        Xp = ops.pad_features(X)
        # one tile-coherent row order per cloud (a function of X alone) serves all three stages: the bandwidth's and the nms
        # membership's sweeps skip the key tiles that provably hold nothing for a 128-row block, the block-sparse iteration kernel
        # runs on the sorted rows
        prep = ops.ms_sparse_prepare(Xp) if (ops.ms_prep_ok(Xp) and ops.MS_SPARSE != "off" and ops._MS_VARIANT == "auto") else None
        if bw is None:
            K = int(quantile * num_samples)
            if num_samples >= N:
                # every row is kept: the statistic does not depend on the order. The reference still consumes one
                # np.random.shuffle per cloud here (mean_shift.py:126-128); `match_rng` replays that (0.15 ms of host time
                # per cloud and pass) for callers that share the global numpy stream with other code
                if self.match_rng:
                    for _ in range(B):
                        self._subset(N, num_samples, "cpu")
                Xs, bprep = Xp, prep
            else:
                # one subset per cloud, drawn cloud by cloud like the reference's per-cloud calls
                sub = torch.stack([self._subset(N, num_samples, X.device) for _ in range(B)])
                Xs, bprep = torch.gather(Xp, 1, sub.unsqueeze(-1).expand(B, num_samples, Xp.shape[2])), None
            bw = ops.ms_bandwidth(Xs.contiguous(), K, 0.003, prep=bprep)
            ops.finite_canary("ms_bandwidth", {"bw": bw}, {"X": Xp})
        new_Xp = ops.ms_iterate(Xp, bw, iterations, prep=prep)
        ops.finite_canary("ms_iterate", {"new_X": new_Xp}, {"X": Xp, "bw": bw})
        labels, ids, n_c, n_l = ops.ms_nms(new_Xp, Xp, bw, prep=prep)
        if ops.FINITE_CANARY:                    # nms: integer outputs -- labels must be valid segment ids, one centre at least
            bad = (labels < 0) | (labels >= N)
            if bool(bad.any()) or bool((n_l <= 0).any()):
                ops.finite_canary("ms_nms", {"labels_invalid": torch.where(bad, float("nan"), 0.0)}, {"new_X": new_Xp, "X": Xp, "bw": bw})
        return new_Xp[:, :, :d], bw, labels, ids, n_c, n_l

    def guard_mean_shift_batch(self, X, quantile, iterations, num_samples=10000, factor=1.2, max_clusters=49,
                               dist=None):
        """Batched form of the script-level guard loop (generate_predictions_aug.py:25-35): every cloud
        whose label count exceeds `max_clusters` is re-run with its own quantile *= factor -- as further passes over
        ONLY those clouds. With `dist` (an initialised torch.distributed; a world of one rank takes the same path) the retry passes
        are collective and spread over all ranks (sednet_hip.shard.balanced_guard_retries); every rank must then call this method.
        -> (labels [B,N] i32, bw [B], n_labels [B] i32 (host), passes [B] (host))."""
        B = X.shape[0]
        q = np.full(B, float(quantile))
        labels = torch.empty(X.shape[:2], dtype=torch.int32, device=X.device)
        bw = torch.empty((B,), dtype=torch.float32, device=X.device)
        n_labels = np.zeros(B, np.int64)
        passes = np.zeros(B, np.int64)

        def run(Xs, qs):
            """one pass over the clouds Xs with per-cloud quantiles qs: clouds sharing a quantile share K -> one launch"""
            lab = torch.empty(Xs.shape[:2], dtype=torch.int32, device=Xs.device)
            b = torch.empty((Xs.shape[0],), dtype=torch.float32, device=Xs.device)
            nl = torch.empty((Xs.shape[0],), dtype=torch.int64, device=Xs.device)
            for qq in np.unique(qs):
                members = np.nonzero(qs == qq)[0]
                whole = members.size == Xs.shape[0]
                sel = None if whole else torch.as_tensor(members, device=Xs.device)
                _, bw_g, lab_g, _, _, nl_g = self.mean_shift_batch(Xs if whole else Xs[sel], num_samples, float(qq),
                                                                   iterations)
                if whole:
                    lab, b, nl = lab_g, bw_g, nl_g.long()
                else:
                    lab[sel], b[sel], nl[sel] = lab_g, bw_g, nl_g.long()
            return lab, b, nl

        distributed = dist is not None and dist.is_initialized()
        todo = np.arange(B)
        first = True
        while distributed or todo.size:
            sel = torch.as_tensor(todo, device=X.device)
            whole = todo.size == B and B > 0
            if first or not distributed:                 # first pass: every rank on its own clouds
                if todo.size:
                    lab_t, bw_t, nl_t = run(X if whole else X[sel], q[todo])
            else:                                        # retry passes: collective, spread over all ranks
                from sednet_hip.shard import balanced_guard_retries
                lab_t, bw_t, nl_t = balanced_guard_retries(X[sel], q[todo], run, dist)
                whole = False
            nxt = []
            if todo.size:
                if whole:
                    labels, bw = lab_t, bw_t
                else:
                    labels[sel], bw[sel] = lab_t, bw_t
                nl = nl_t.cpu().numpy()          # the one D->H sync per pass (reference: :31)
                for b_, n in zip(todo, nl):
                    n_labels[b_] = n
                    passes[b_] += 1
                    if n > max_clusters:
                        q[b_] *= factor
                        nxt.append(b_)
            todo = np.array(nxt, dtype=np.int64)
            first = False
            if distributed:
                more = torch.tensor([todo.size], dtype=torch.int64,
                                    device=X.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(more)
                if int(more.item()) == 0:
                    break
        return labels, bw, n_labels, passes

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _check_kernel(kernel_type):
        if kernel_type != "gaussian":
            raise NotImplementedError("only the gaussian kernel is on the HIP path (reference callers use no other)")

    @staticmethod
    def _subset(N, num_samples, device):
        L = np.arange(N)
        np.random.shuffle(L)                      # same RNG consumption as mean_shift.py:126-128
        return torch.as_tensor(L[0:num_samples], device=device)

    def _bandwidth_padded(self, Xp, num_samples, quantile, min_bw=0.003):
        N = Xp.shape[0]
        L = self._subset(N, num_samples, Xp.device)
        # when every row is kept the statistic is permutation invariant: skip the gather
        Xs = Xp if num_samples >= N else Xp[L].contiguous()
        K = int(quantile * num_samples)
        return ops.ms_bandwidth(Xs[None], K, min_bw)[0]
