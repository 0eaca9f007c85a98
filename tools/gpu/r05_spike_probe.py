import sys, tempfile
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "revisit-bpr_amd"))
from revisit_bpr.datasets import synthetic
from revisit_bpr.fast import StreamTrainer
from revisit_bpr.models import BPR
from revisit_bpr.models.bpr import MF
data = synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2, median_per_user=37, min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0, cache_dir=tempfile.gettempdir())
t = {k: torch.from_numpy(getattr(data, k)).cuda() for k in ("users", "items", "indptr", "indices")}
torch.manual_seed(13)
model = BPR(fuse_forward=True, reg_alphas={"user": 0.0016, "item": 0.0001, "neg": 0.00375}, logits_model=MF(torch.nn.Embedding(data.num_users, 128, padding_idx=0), torch.nn.Embedding(data.num_items, 128, padding_idx=0))).cuda()
tr = StreamTrainer(model, t["users"], t["items"], t["indptr"], t["indices"], lr=0.05, sampler="adaptive", adaptive_p=0.01, batch_size=256, seed=1)
tr.train_epoch()
Q = model.logits_model.get_features()["item"].data.cpu().numpy()
cnt = np.bincount(data.items, minlength=data.num_items)
for f in (0, 5, 77):
    col = Q[:, f]
    s = np.sort(col[1:])
    print(f"col {f}: sd {col[1:].std():.5f} mean {col[1:].mean():.5f} min {s[0]:.4f} max {s[-1]:.4f} q01 {s[len(s)//100]:.5f} q25 {s[len(s)//4]:.5f} q50 {s[len(s)//2]:.5f} q75 {s[3*len(s)//4]:.5f} q99 {s[-len(s)//100]:.5f}")
    gaps = np.diff(s)
    # densest window of 128 consecutive sorted keys
    w = s[128:] - s[:-128]
    i = int(np.argmin(w))
    print(f"   densest 128 keys: span {w[i]:.3e} around {s[i+64]:.6f}; unique values in it {len(np.unique(s[i:i+128]))}; median gap overall {np.median(gaps):.3e}")
    idx = np.argsort(col)[i+1:i+129]
    print(f"   their training counts: median {np.median(cnt[idx])}, max {cnt[idx].max()}; share of items with zero training positives overall {(cnt==0).mean():.3f}")
np.save(str(ROOT / "gpurun_out" / "r05_spike_cols.npy"), Q[:, [0, 5, 77, 100]])
