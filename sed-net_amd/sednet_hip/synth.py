"""Synthetic inputs for tests and bench.py (no datasets or checkpoints exist offline).

* `synthetic_cloud(seed, N)`   -- CAD-like cloud of analytic primitives with unit normals,
  centred / unit-scaled / PCA-aligned the way the reference's loader does it
  (/root/reference/src/dataset_segments.py:376-379, :400-402, :412-417).
* `closed_form_state_dict(k)`   -- deterministic SED-Net weights keyed like the reference
  state-dict; closed-form (no RNG state) so every consumer gets identical
  arrays regardless of library versions.
* `clustered_embedding(...)`    -- unit-norm embeddings with well separated clusters for the
  mean-shift stage in isolation (random-weight networks collapse to one cluster).
"""
import zlib

import numpy as np

F32 = np.float32

PLANE, CONE, CYLINDER, SPHERE = 1, 3, 4, 5


# ---------------------------------------------------------------------------------------------
# clouds
# ---------------------------------------------------------------------------------------------

def _frame(rng):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    t = np.cross(a, [1.0, 0, 0] if abs(a[0]) < 0.9 else [0, 1.0, 0])
    t /= np.linalg.norm(t)
    return a, t, np.cross(a, t)


def sample_primitive(kind, n, rng):
    """n points + unit normals on one analytic primitive patch."""
    a, u, v = _frame(rng)
    c = rng.uniform(-0.6, 0.6, size=3)
    if kind == PLANE:
        s, t = rng.uniform(-0.5, 0.5, size=(2, n)) * rng.uniform(0.3, 1.0, size=(2, 1))
        p = c + s[:, None] * u + t[:, None] * v
        nr = np.broadcast_to(a, p.shape).copy()
    elif kind == SPHERE:
        r = rng.uniform(0.15, 0.45)
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        p, nr = c + r * d, d
    elif kind == CYLINDER:
        r = rng.uniform(0.1, 0.35)
        h = rng.uniform(-0.5, 0.5, size=n) * rng.uniform(0.4, 1.2)
        phi = rng.uniform(0, 2 * np.pi, size=n)
        d = np.cos(phi)[:, None] * u + np.sin(phi)[:, None] * v
        p, nr = c + h[:, None] * a + r * d, d
    elif kind == CONE:
        th = rng.uniform(0.25, 0.9)
        h = rng.uniform(0.15, 0.8, size=n)
        phi = rng.uniform(0, 2 * np.pi, size=n)
        d = np.cos(phi)[:, None] * u + np.sin(phi)[:, None] * v
        p = c + h[:, None] * a + (h * np.tan(th))[:, None] * d
        nr = np.cos(th) * d - np.sin(th) * a           # outward normal of the cone surface
    else:
        raise ValueError(kind)
    return p, nr


def synthetic_cloud(seed, N=10000, n_prims=None, noise=0.0):
    """-> points f32[N,3], normals f32[N,3], labels i64[N], types i64[N]."""
    rng = np.random.default_rng(seed)
    if n_prims is None:
        n_prims = int(rng.integers(8, 17))
    kinds = rng.choice([PLANE, CONE, CYLINDER, SPHERE], size=n_prims)
    share = rng.uniform(0.5, 1.5, size=n_prims)
    counts = np.maximum((share / share.sum() * N).astype(int), 24)
    counts[-1] += N - counts.sum()
    if counts[-1] < 24:                       # rebalance so every patch keeps >= 24 points
        counts = np.full(n_prims, N // n_prims)
        counts[-1] += N - counts.sum()
    P, Nr, L, T = [], [], [], []
    for i, (kd, cnt) in enumerate(zip(kinds, counts)):
        p, nr = sample_primitive(int(kd), int(cnt), rng)
        P.append(p); Nr.append(nr); L.append(np.full(cnt, i)); T.append(np.full(cnt, kd))
    P, Nr = np.concatenate(P), np.concatenate(Nr)
    L, T = np.concatenate(L), np.concatenate(T)
    if noise > 0:
        P = P + rng.normal(scale=noise, size=P.shape)
    perm = rng.permutation(N)
    P, Nr, L, T = P[perm], Nr[perm], L[perm], T[perm]
    P = P - P.mean(0)                                           # dataset_segments.py:376-379
    P = P / (np.max(P.max(0) - P.min(0)) + 1e-12)               # :400-402
    w, V = np.linalg.eigh(P.T @ P)                              # :412-417 smallest axis -> x
    R = V[:, [0, 1, 2]].T
    if np.linalg.det(R) < 0:
        R[2] = -R[2]
    P, Nr = P @ R.T, Nr @ R.T
    return P.astype(F32), Nr.astype(F32), L.astype(np.int64), T.astype(np.int64)


def batch_clouds(B, N=10000, seed0=1234):
    """-> x f32[B,6,N] (channel-major as SEDNet.forward takes it), labels [B,N], types [B,N]."""
    xs, ls, ts = [], [], []
    for b in range(B):
        p, n, l, t = synthetic_cloud(seed0 + b, N)
        xs.append(np.concatenate([p, n], 1).T)
        ls.append(l); ts.append(t)
    return np.ascontiguousarray(np.stack(xs)), np.stack(ls), np.stack(ts)


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------

STATE_SHAPES = {
    "encoder.bn1.weight": (64,), "encoder.bn1.bias": (64,),
    "encoder.bn2.weight": (64,), "encoder.bn2.bias": (64,),
    "encoder.bn3.weight": (128,), "encoder.bn3.bias": (128,),
    "encoder.bn4.weight": (256,), "encoder.bn4.bias": (256,),
    "encoder.bn5.weight": (1024,), "encoder.bn5.bias": (1024,),
    "encoder.conv1.0.weight": (64, 12, 1, 1),
    "encoder.conv1.1.weight": (64,), "encoder.conv1.1.bias": (64,),
    "encoder.conv2.0.weight": (64, 128, 1, 1),
    "encoder.conv2.1.weight": (64,), "encoder.conv2.1.bias": (64,),
    "encoder.conv3.0.weight": (128, 128, 1, 1),
    "encoder.conv3.1.weight": (128,), "encoder.conv3.1.bias": (128,),
    "encoder.mlp1.weight": (1024, 256, 1), "encoder.mlp1.bias": (1024,),
    "encoder.bnmlp1.weight": (1024,), "encoder.bnmlp1.bias": (1024,),
    "conv1.weight": (512, 1280, 1), "conv1.bias": (512,),
    "bn1.weight": (512,), "bn1.bias": (512,),
    "conv2.weight": (256, 512, 1), "conv2.bias": (256,),
    "bn2.weight": (256,), "bn2.bias": (256,),
    "edge_module.0.weight": (128, 256, 1), "edge_module.0.bias": (128,),
    "edge_module.1.weight": (128,), "edge_module.1.bias": (128,),
    "edge_module.2.weight": (2, 128, 1), "edge_module.2.bias": (2,),
    "asis.0.weight": (256, 256, 1), "asis.0.bias": (256,),
    "asis.1.weight": (256,), "asis.1.bias": (256,),
    "mlp_seg_prob1.weight": (256, 256, 1), "mlp_seg_prob1.bias": (256,),
    "mlp_seg_prob2.weight": (128, 256, 1), "mlp_seg_prob2.bias": (128,),
    "bn_seg_prob1.weight": (256,), "bn_seg_prob1.bias": (256,),
    "mlp_prim_prob1.weight": (256, 256, 1), "mlp_prim_prob1.bias": (256,),
    "mlp_prim_prob2.weight": (6, 256, 1), "mlp_prim_prob2.bias": (6,),
    "bn_prim_prob1.weight": (256,), "bn_prim_prob1.bias": (256,),
    "prim_encoding.0.weight": (256, 8, 1), "prim_encoding.0.bias": (256,),
}

# encoder.bnK aliases encoder.convK.1 in the reference (SEDNet.py:31-45): same arrays.
_ALIASES = {"encoder.conv1.1": "encoder.bn1", "encoder.conv2.1": "encoder.bn2", "encoder.conv3.1": "encoder.bn3"}


def _closed_form(key, shape, salt):
    n = int(np.prod(shape))
    h = (zlib.crc32(key.encode()) + 7919 * salt) % 100003
    i = np.arange(n, dtype=np.float64)
    base = np.sin(i * 12.9898 + h * 0.618034) * 43758.5453
    u = base - np.floor(base)                                   # in [0,1)
    leaf = key.rsplit(".", 1)[-1]
    is_norm = len(shape) == 1 and leaf == "weight"
    if is_norm:                                                 # GroupNorm gamma: ~1, a few negative
        v = 0.6 + 0.8 * u
        v[(np.arange(n) % 11) == 3] *= -1.0
    elif leaf == "bias":
        v = (u - 0.5) * 0.2
    else:
        fan_in = int(np.prod(shape[1:]))
        v = (u - 0.5) * 2.0 * np.sqrt(3.0 / fan_in)
    return v.reshape(shape).astype(F32)


def closed_form_state_dict(salt=0):
    """Deterministic weights for one SEDNet (`salt` distinguishes the type / instance models)."""
    sd = {}
    for key, shape in STATE_SHAPES.items():
        mod = key.rsplit(".", 1)[0]
        src = key.replace(mod, _ALIASES[mod]) if mod in _ALIASES else key
        sd[key] = _closed_form(src, shape, salt)
    return sd


_TRAINED = {}


def trained_weights_path():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(os.path.dirname(here)), "tests", "golden", "w_trained.npz")


def trained_state_dict(role):
    """State dict of a network TRAINED by the reference's own training step on synthetic clouds (tests/golden/train_weights.py
    ran /root/reference/train_sed_net.py:233-283's loss under the CPU shim; the arrays are a data fixture). role: "type" or
    "inst" -- two snapshots of the run, like the script's two checkpoints (generate_predictions_aug.py:142-167). Unlike the
    closed-form weights these separate the segments of a synthetic cloud: >= 3 primitive types and >= 8 mean-shift clusters per
    cloud, which is what the end-to-end fixtures (f_e2e, f_10k) and bench.py's headline need."""
    if role not in ("type", "inst"):
        raise ValueError(role)
    if not _TRAINED:
        with np.load(trained_weights_path()) as z:
            for k in z.files:
                r, name = k.split("/", 1)
                _TRAINED.setdefault(r, {})[name] = z[k]
    return dict(_TRAINED[role])


# ---------------------------------------------------------------------------------------------
# embeddings for the clustering stage in isolation
# ---------------------------------------------------------------------------------------------

def clustered_embedding(N=10000, d=128, n_clusters=12, sigma=0.01, seed=0):
    """Unit-norm rows around `n_clusters` random unit centres (sigma = per-coordinate noise)."""
    rng = np.random.default_rng(seed)
    C = rng.normal(size=(n_clusters, d))
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    assign = rng.integers(0, n_clusters, size=N)
    X = C[assign] + rng.normal(scale=sigma, size=(N, d))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(F32), assign.astype(np.int64)


def realistic_embedding(N=10000, d=128, n_clusters=14, sigma=0.02, bridge=0.04, seed=0):
    """Embedding of a "trained-network-like" cloud: unit rows around `n_clusters` centres of UNEQUAL size, two of the
    centres close to each other (0.35 rad), and a fraction `bridge` of the points spread along the arc between
    neighbouring centres (the points whose cluster is decided by the last bits of the mean-shift arithmetic).
    -> (X [N,d] fp32, nominal assignment [N])."""
    rng = np.random.default_rng(seed)
    C = rng.normal(size=(n_clusters, d))
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    t = rng.normal(size=d); t -= t.dot(C[0]) * C[0]; t /= np.linalg.norm(t)
    C[1] = np.cos(0.35) * C[0] + np.sin(0.35) * t                     # a close pair
    sizes = rng.dirichlet(np.full(n_clusters, 2.0))
    assign = rng.choice(n_clusters, size=N, p=sizes)
    X = C[assign] + rng.normal(scale=sigma, size=(N, d))
    nb = int(bridge * N)
    rows = rng.choice(N, nb, replace=False)
    other = (assign[rows] + 1 + rng.integers(0, n_clusters - 1, size=nb)) % n_clusters
    lam = rng.uniform(0.0, 0.5, size=(nb, 1))
    X[rows] = (1 - lam) * C[assign[rows]] + lam * C[other] + rng.normal(scale=sigma, size=(nb, d))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(F32), assign.astype(np.int64)


def planted_embedding(labels, d=128, sigma=0.01, seed=0, guard_clouds=()):
    """Per-point embedding that carries the segment structure of `labels` [B,N] (numpy / torch ints): unit centre per
    (cloud, segment) + sigma noise per coordinate, built on the device of the returned tensor (cuda if available).
    Clouds listed in `guard_clouds` get instead 60 tight clusters in 30 close pairs (unrelated to the labels): more
    than 49 clusters at the script's quantile 0.015, the pairs merge one guard retry later (x1.2) -- the case
    generate_predictions_aug.py:25-35 exists for. -> (X [B,N,d] fp32 unit rows, planted [B,N] int64: the partition the
    clustering must return; for guard clouds the pair index)."""
    import torch
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    L = torch.as_tensor(np.asarray(labels)).long().to(dev)
    B, N = L.shape
    g = torch.Generator(device=dev).manual_seed(seed)
    S = int(L.max()) + 1
    C = torch.nn.functional.normalize(torch.randn(B, max(S, 60), d, generator=g, device=dev), dim=2)
    planted = L.clone()
    for b in guard_clouds:
        twin = C[b, :30] + 0.04 * torch.randn(30, d, generator=g, device=dev)
        C[b, 30:60] = torch.nn.functional.normalize(twin, dim=1)
        a = torch.arange(N, device=dev) % 60
        planted[b] = a % 30
        L = L.clone(); L[b] = a
    X = torch.gather(C, 1, L[:, :, None].expand(B, N, d))
    sig = torch.full((B, 1, 1), float(sigma), device=dev)
    for b in guard_clouds:
        sig[b] = 0.002
    X = X + sig * torch.randn(B, N, d, generator=g, device=dev)
    return torch.nn.functional.normalize(X, dim=2).float().contiguous(), planted
