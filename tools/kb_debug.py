import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from msmdfusion_amd import kernels as K
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for cin, cout, n in [(128, 192, 1500), (128, 128, 1500), (64, 64, 1500)]:
        f = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.1
        from msmdfusion_amd import synthetic as S
        idx = torch.from_numpy(S.random_voxel_indices(n, 2, [11, 64, 64], seed=cin + cout)).to(dev)
        n = idx.shape[0]
        f = torch.randn(n, cin, device=dev)
        nbr = K.rulebook_subm(idx, 2, [11, 64, 64], 3)
        ws = K.pack_weight_split(w, 3)
        outs = [K.conv_forward_split(f, ws, nbr, n, cout, 3) for _ in range(3)]
        torch.save([o.cpu() for o in outs], "/tmp/kb_%s_%d_%d.pt" % (sys.argv[1], cin, cout))
        print(sys.argv[1], cin, cout, n, "self-consistent:", torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]))
else:
    import torch
    for kb in ("1", "0"):
        subprocess.run([sys.executable, __file__, kb], env=dict(os.environ, MSMD_SPLIT_KB=kb))
    for cin, cout, n in [(128, 192, 1500), (128, 128, 1500), (64, 64, 1500)]:
        a = torch.load("/tmp/kb_1_%d_%d.pt" % (cin, cout))[0]
        b = torch.load("/tmp/kb_0_%d_%d.pt" % (cin, cout))[0]
        d = (a - b).abs()
        bad = (d > 1e-3).nonzero()
        print(cin, cout, n, "max diff %.3e" % d.max().item(), "bad elems", bad.shape[0],
              "rows", sorted(set(bad[:, 0].tolist()))[:10], "cols", sorted(set(bad[:, 1].tolist()))[:6], "...")
