import torch
x = torch.empty((1000000, 5408), dtype=torch.float64, device="cuda")
for name, fn in (("zero_", lambda: x.zero_()), ("fill_(1.5)", lambda: x.fill_(1.5))):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print("%-12s median %.2f ms -> %.0f GB/s written" % (name, ts[3], x.numel() * 8 / ts[3] / 1e6))
