import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from msmdfusion_amd import kernels as K
dev = torch.device('cuda:0')
for n in (8000, 16000, 20000, 24000, 40000):
    rng = np.random.RandomState(n)
    xyz = torch.from_numpy(np.stack([rng.randint(0, 41, (2, n)), rng.randint(0, 1440, (2, n)), rng.randint(0, 1440, (2, n))], -1).astype(np.float32)).to(dev)
    for _ in range(2): K.furthest_point_sample(xyz, 2048)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); K.furthest_point_sample(xyz, 2048); e.record(); torch.cuda.synchronize()
    print("fps n=%d m=2048 b=2: %.2f ms" % (n, s.elapsed_time(e)))
