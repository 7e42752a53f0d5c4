import time, torch
x = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
y = torch.empty_like(x)
for n in (100, 1000, 3000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        torch.mm(x, x, out=y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d launches: host %.1f ms, GPU done %.1f ms" % (n, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
