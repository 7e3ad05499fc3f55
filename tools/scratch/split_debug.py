import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmdfusion_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
for cin, cout, n, kvol in [(32, 32, 40, 1), (32, 32, 200, 3), (128, 128, 300, 27), (64, 64, 300, 27)]:
    f = torch.randn(n, cin, device=dev)
    w = torch.randn(kvol, cin, cout, device=dev) * 0.1
    nbr = torch.randint(-n, n, (kvol, n), device=dev, dtype=torch.int32).clamp_(min=-1)
    nbr[0] = torch.arange(n, device=dev, dtype=torch.int32)
    ref = torch.zeros(n, cout, dtype=torch.float64, device=dev)
    for k in range(kvol):
        m = nbr[k] >= 0
        ref[m] += f.double()[nbr[k][m].long()] @ w.double()[k]
    for planes in (3, 1):
        ws = K.pack_weight_split(w, planes)
        o = K.conv_forward_split(f, ws, nbr, n, cout, planes).double()
        err = (o - ref).abs()
        print(cin, cout, n, kvol, "planes", planes, "max err %.3e" % err.max().item(), "ref max %.2f" % ref.abs().max().item(),
              "bad rows", (err.max(1)[0] > 1e-2).sum().item(), "bad cols", (err.max(0)[0] > 1e-2).sum().item())
        if planes == 3 and err.max() > 1e-3:
            r = int(err.max(1)[0].argmax()); print(" row", r, "got", o[r, :6].tolist(), "ref", ref[r, :6].tolist())
