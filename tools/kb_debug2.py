import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
dev = torch.device("cuda:0")
cin, cout = 128, 192
shape = [11, 64, 64]
idx = S.random_voxel_indices(1500, 2, shape, seed=cin + cout)
n = idx.shape[0]
rng = np.random.RandomState(cin * 1000 + cout)
f = rng.randn(n, cin).astype(np.float32)
w = (rng.randn(27, cin, cout) / np.sqrt(27 * cin)).astype(np.float32)
nbr = K.rulebook_subm(torch.from_numpy(idx).to(dev), 2, shape, 3)
fd, wd = torch.from_numpy(f).to(dev), torch.from_numpy(w).to(dev)
ws = K.pack_weight_split(wd, 3)
out = K.conv_forward_split(fd, ws, nbr, n, cout, 3)
ref = torch.zeros(n, cout, dtype=torch.float64, device=dev)
for k in range(27):
    m = nbr[k] >= 0
    ref[m] += fd.double()[nbr[k][m].long()] @ wd.double()[k]
d = (out.double() - ref).abs()
bad = (d > 1e-3).nonzero()
print("n", n, "bad", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:8], "max", d.max().item())
if bad.shape[0]:
    r = int(bad[0, 0]); print("row", r, "tile", r // 128, "in-tile", r % 128, "mask of row", [int(nbr[k][r] >= 0) for k in range(27)])
    print("got", out[r, :4].tolist(), "ref", ref[r, :4].tolist(), "partial check: ratio", (out[r, :8].double() / ref[r, :8]).tolist())
