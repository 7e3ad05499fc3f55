import glob, os, torch
bad = 0
for a in sorted(glob.glob("gpurun_out/wv/v0/*.pt")):
    b = a.replace("/v0/", "/v1/")
    x, y = torch.load(a), torch.load(b)
    same = torch.equal(x, y)
    bad += not same
    print(os.path.basename(a), "bit-identical" if same else "DIFF max %g" % (x - y).abs().max().item())
print("all identical" if not bad else "%d differ" % bad)
