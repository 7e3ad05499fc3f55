import torch, time
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n
for n in (75000, 150000):
    k32 = torch.randint(-2**31, 2**31 - 1, (n,), device=dev, dtype=torch.int64).int()
    k16 = torch.randint(-2**15, 2**15 - 1, (n,), device=dev, dtype=torch.int64).short()
    k8 = torch.randint(0, 255, (n,), device=dev, dtype=torch.int64).to(torch.uint8)
    print(n, "int32 sort %.0f us" % timed(lambda: torch.sort(k32, stable=False)),
          "int16 sort %.0f us" % timed(lambda: torch.sort(k16, stable=False)),
          "uint8 sort %.0f us" % timed(lambda: torch.sort(k8, stable=False)),
          "int16 argsort stable %.0f us" % timed(lambda: torch.sort(k16, stable=True)))
