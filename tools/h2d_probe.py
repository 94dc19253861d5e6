import torch, time
dev = 'cuda:0'
A = torch.randn(8192, 8192, device=dev)
def busy():
    for _ in range(4): torch.mm(A, A)
for name, fn in (("torch.tensor([x], device)", lambda: torch.tensor([3.0], dtype=torch.float32, device=dev)),
                 ("torch.tensor(tuple of 12, device)", lambda: torch.tensor((1.0,) * 12, dtype=torch.float32, device=dev)),
                 ("torch.full", lambda: torch.full((1,), 3.0, dtype=torch.float32, device=dev)),
                 ("pinned.to(device, non_blocking)", lambda: torch.tensor([3.0]).pin_memory().to(dev, non_blocking=True))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    busy(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:40s}: host {1e3*(t1-t0):7.3f} ms with the stream busy (drain took {1e3*(t2-t1):.1f} ms more)")
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
    print(f"{'':40s}  host {1e3*(t1-t0):7.3f} ms with the stream idle")
