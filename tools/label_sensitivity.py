"""Which mean-shift schedule / arithmetic moves which labels on a trained-network embedding (bench cloud `seed`):
python tools/label_sensitivity.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from conftest import label_agreement
from sednet_hip import ops, synth
from src.mean_shift import MeanShift
from src.SEDNet import SEDNet
from test_gpu_mean_shift import set_schedule, reset_schedule

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1235
tag = {1234: "", 1235: "c1_"}.get(seed)
g = np.load(os.path.join(ROOT, "tests", "golden", "f_10k.npz"))
p, n, gl, gt = synth.synthetic_cloud(seed, 10000)
x = torch.from_numpy(np.concatenate([p, n], 1).T[None].astype(np.float32)).cuda()
m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6, combine_label_prim=True,
           edge_module=True, late_fusion=True, nn_nb=20)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.trained_state_dict("inst").items()})
m = m.cuda().eval()
with torch.no_grad():
    emb = m(x, None, False)[0][0].T.contiguous()
X = torch.nn.functional.normalize(emb, p=2, dim=1)
ms = MeanShift()
res, rows = {}, {}
for v in ("batched", "splitk", "chunked", "f16/1", "f16", "f16c/1", "f16c", "sparse/1", "sparse"):
    set_schedule(v)
    nx, cen, bw, ids = ms.mean_shift(X, 10000, 0.015, 50)
    res[v] = ids.cpu().numpy(); rows[v] = nx.cpu().numpy()
    reset_schedule()
print("bw", float(bw), "clusters", {v: len(np.unique(r)) for v, r in res.items()})
if tag is not None:
    ref, mar = g[tag + "labels"], g[tag + "label_margin"].astype(np.float32)
    for v, r in res.items():
        a = label_agreement(r, ref, mar, 5e-3)
        print(f"{v:9s} vs reference: rate {a['rate']:.5f} differ {a['mismatches'].size:3d} undecided {a['undecided'].size}  "
              f"max ref margin of differing {mar[a['mismatches']].max() if a['mismatches'].size else 0:.2e}")
names = list(res)
print("pairwise differing points:")
for i, a in enumerate(names):
    print(f"{a:9s}", " ".join(f"{label_agreement(res[a], res[b])['mismatches'].size:4d}" for b in names))
print("max |rows - batched rows|:", {v: float(np.abs(rows[v] - rows['batched']).max()) for v in names})
bad = label_agreement(res["f16/1"], res["batched"])["mismatches"]
if bad.size:
    d = np.abs(rows["f16/1"][bad] - rows["batched"][bad]).max(1)
    print("rows of the differing points, max |f16 - batched|:", np.sort(d)[::-1][:10])
    # where do those rows sit after 50 iterations: distance to the nearest two selected centres (batched)
