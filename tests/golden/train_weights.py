"""Train the REFERENCE network with the REFERENCE's own training step on synthetic clouds (CPU, build container only)
and store the resulting state dicts as a data fixture: tests/golden/w_trained.npz.

Why ("what's weak" 1): the closed-form weights of sednet_hip.synth collapse every cloud to one
primitive type and one mean-shift cluster, so no end-to-end fixture ever tested labels / types with backbone error
flowing into a multi-cluster clustering stage. A network that has been trained for a few hundred steps separates the
segments of the synthetic clouds (>= 3 types, >= 8 clusters per cloud).

What runs: /root/reference/src/SEDNet.py (model), src/segment_loss.py (triplet + smoothed CE), src/My_edge_loss.py
(edge CE + edge-embedding loss), combined exactly like train_sed_net.py:233-283 (AdamW, weight decay 0.002 as in
configs/config_SEDNet_normal.yml:40,49; the learning rate is raised from the config's 1e-4 to 1e-3 because only a few
hundred steps are affordable on 8 CPU cores). Inputs: sednet_hip.synth.synthetic_cloud (seeds 50000+), edge labels =
points with a differently-labelled point among their 8 nearest neighbours.

Two snapshots are kept, like the script's two checkpoints (generate_predictions_aug.py:142-167: one model for the
types, one for the instance embedding): "type" = the state after `--steps-type` steps, "inst" = after `--steps` steps.

Usage:  python tests/golden/train_weights.py [--steps 400] [--steps-type 300] [--out tests/golden/w_trained.npz]
"""
import argparse
import importlib.util
import os
import sys
import time
import warnings

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "sed-net_amd", "sednet_hip", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

F32 = np.float32


def edge_labels(p, l, k=8):
    """Per-point boundary flag: a point whose k nearest neighbours (xyz) hold a different segment label."""
    P = torch.from_numpy(p)
    out = np.zeros(p.shape[0], np.int64)
    for s in range(0, p.shape[0], 2048):
        d = torch.cdist(P[s:s + 2048], P)
        nn_ = d.topk(k + 1, largest=False).indices.numpy()
        out[s:s + 2048] = (l[nn_] != l[s:s + 2048, None]).any(1)
    return out


def batch(step, B, N):
    xs, ls, ts, es = [], [], [], []
    for b in range(B):
        p, n, l, t = synth.synthetic_cloud(50000 + step * 16 + b, N)
        xs.append(np.concatenate([p, n], 1).T)
        ls.append(l); ts.append(t); es.append(edge_labels(p, l))
    return (torch.from_numpy(np.stack(xs).astype(F32)), np.stack(ls), torch.from_numpy(np.stack(ts)),
            torch.from_numpy(np.stack(es)))


def evaluate(model, seed, N, k):
    """Types / clusters the reference's own inference flow finds on one cloud (generate_predictions_aug.py:221-236,
    365, 380-382 with guard_mean_shift's first pass only)."""
    from src.mean_shift import MeanShift
    p, n, l, t = synth.synthetic_cloud(seed, N)
    x = torch.from_numpy(np.concatenate([p, n], 1).T[None].astype(F32))
    model.eval()
    with torch.no_grad():
        emb, logp, _, _ = model(x, None, False)
    model.train()
    ty = logp[0].argmax(0).numpy()
    X = torch.nn.functional.normalize(emb[0].T, p=2, dim=1)
    np.random.seed(0)
    _, _, bw, ids = MeanShift().mean_shift(X, 10000, 0.015, 50)
    return (int(np.unique(ty).shape[0]), float((ty == t).mean()), int(torch.unique(ids).shape[0]),
            int(np.unique(l).shape[0]), float(bw))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--steps-type", type=int, default=300)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(HERE, "w_trained.npz"))
    ap.add_argument("--resume", default="")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)

    from src.SEDNet import SEDNet
    from src.segment_loss import EmbeddingLoss, LabelSmoothingLoss
    from src.My_edge_loss import edge_cls_loss, compute_edge_embedding_loss

    torch.manual_seed(2024)
    np.random.seed(2024)
    Loss = EmbeddingLoss(margin=1.0, if_mean_shift=False)
    smooth = LabelSmoothingLoss(smoothing=0.025)
    model = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, loss_function=Loss.triplet_loss,
                   mode=5, num_channels=6, combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=a.k)
    opt = torch.optim.AdamW(model.parameters(), lr=a.lr, weight_decay=0.002)
    model.train()
    start = 0
    snaps = {}
    if a.resume:
        ck = torch.load(a.resume)
        model.load_state_dict(ck["model"]); opt.load_state_dict(ck["opt"]); start = ck["step"]; snaps = ck.get("snaps", {})
    sizes = [(8, 1024), (4, 2048), (2, 4096), (8, 1024), (4, 2048), (1, 10000)]
    t0 = time.time()
    for step in range(start, a.steps):
        B, N = sizes[step % len(sizes)]
        if step >= a.steps - 40:                      # finish at the benchmarked cloud size
            B, N = 1, 10000
        x, labels, prims, edges = batch(step, B, N)
        opt.zero_grad()
        emb, logp, _, edges_pred = model(points=x)
        embed_loss = torch.mean(Loss.triplet_loss(emb, labels))
        edge_loss = edge_cls_loss(edges_pred, edges, torch.ones_like(edges, dtype=torch.float32))
        p_loss = smooth(logp.transpose(1, 2).contiguous().view(-1, 6), prims.contiguous().view(-1))
        ee = compute_edge_embedding_loss(edges_pred=edges_pred, pred_feat=emb, gt_label=torch.from_numpy(labels),
                                         use_type=True, primitives=prims, primitives_log_prob=logp)
        loss = embed_loss + p_loss + edge_loss + 0.25 * ee          # train_sed_net.py:266
        loss.backward()
        opt.step()
        print(f"step {step} B{B} N{N} loss {loss.item():.4f} emb {embed_loss.item():.4f} prim {p_loss.item():.4f} "
              f"edge {edge_loss.item():.4f} ee {ee.item():.4f}  {time.time() - t0:.0f}s", flush=True)
        if step + 1 == a.steps_type:
            snaps["type"] = {k_: v.detach().clone() for k_, v in model.state_dict().items()}
        if (step + 1) % 50 == 0 or step + 1 == a.steps:
            print("  eval N=1024 (types found, type acc, clusters, true segments, bw):", evaluate(model, 21, 1024, a.k),
                  flush=True)
            torch.save({"model": model.state_dict(), "opt": opt.state_dict(), "step": step + 1, "snaps": snaps},
                       "/tmp/w_trained_ckpt.pt")
    snaps["inst"] = {k_: v.detach().clone() for k_, v in model.state_dict().items()}
    print("  eval N=10000:", evaluate(model, 1234, 10000, a.k), flush=True)
    out = {}
    for role, sd in snaps.items():
        for k_, v in sd.items():
            if k_.startswith("pos_enc"):
                continue
            out[f"{role}/{k_}"] = v.numpy()
    np.savez_compressed(a.out, **out)
    print("wrote", a.out, os.path.getsize(a.out) / 1e6, "MB")


if __name__ == "__main__":
    main()
