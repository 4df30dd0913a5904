import torch, time
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(3): x.add_(1.0)
    torch.cuda.synchronize()
    reps = max(5, 4096 // mb)
    t0 = time.perf_counter()
    for _ in range(reps): x.add_(1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{mb:5d} MB working set: read+write {2 * mb / 1024 / dt / 1000:.2f} TB/s ({dt * 1e6:.0f} us per pass)", flush=True)
    del x
