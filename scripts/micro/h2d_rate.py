import torch, time
for mb in (4, 8, 12, 48):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print("H2D %3d MB pinned: %.3f ms = %.1f GB/s" % (mb, dt * 1e3, mb / 1024 / dt))
