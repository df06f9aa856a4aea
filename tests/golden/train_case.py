"""Seeded inputs and the gradient digest of the training fixture f_train.npz -- shared by the generator (which runs
the reference) and by the tests (which rebuild the same inputs without it)."""
import numpy as np

F32 = np.float32


def train_case(synth, N=384, B=2, seed0=300):
    """Inputs of the training fixture, rebuilt identically by the tests (seeded, no reference needed)."""
    xs, labels, types = [], [], []
    for b in range(B):
        p, n, l, ty = synth.synthetic_cloud(seed0 + b, N, n_prims=5)
        xs.append(np.concatenate([p, n], 1).T)
        labels.append(l)
        types.append(ty)
    rng = np.random.default_rng(seed0 + 99)
    edges = (rng.random((B, N)) < 0.2).astype(np.int64)
    edges_w = (rng.random((B, N)) < 0.7).astype(F32) * rng.uniform(0.5, 2.0, size=(B, N)).astype(F32)
    cot = {"emb": rng.normal(size=(B, 128, N)).astype(F32), "logp": rng.normal(size=(B, 6, N)).astype(F32),
           "edges": rng.normal(size=(B, 2, N)).astype(F32)}
    return (np.stack(xs).astype(F32), np.stack(labels), np.stack(types), edges, edges_w, cot)


def grad_digest(named_grads):
    """Full gradient for small parameters; for large ones 2048 evenly spaced entries + sum + L2 norm."""
    out = {}
    for name, g in named_grads:
        g = g.detach().numpy().reshape(-1)
        if g.size <= 4096:
            out["g:" + name] = g.astype(F32)
        else:
            sel = np.linspace(0, g.size - 1, 2048).astype(np.int64)
            out["g:" + name] = g[sel].astype(F32)
        out["n:" + name] = np.array([g.astype(np.float64).sum(), np.linalg.norm(g.astype(np.float64))])
    return out
