import time, torch
torch.cuda.init()
s = torch.cuda.Stream()
for mb in (1, 4, 15, 60, 118):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        with torch.cuda.stream(s):
            t0 = time.perf_counter()
            d.copy_(h, non_blocking=True)
            t1 = time.perf_counter()
        s.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort()
    print(f"{mb} MB: enqueue {ts[10][0]*1e6:.0f} us, complete {ts[10][1]*1e6:.0f} us = {n/ts[10][1]/1e9:.1f} GB/s")
