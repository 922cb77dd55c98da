"""Host <-> device copy rates of this box (torch, pinned and pageable, 24 MB and 256 MB): the ceiling of solve_load / solve_fetch and of the LHS upload."""
import time, torch
for mb in (24, 256):
    n = mb * (1 << 20) // 8
    hp = torch.empty(n, dtype=torch.float64).pin_memory(); hq = torch.empty(n, dtype=torch.float64)
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    for name, src, dst in (("pinned H2D", hp, d), ("pinned D2H", d, hp), ("pageable H2D", hq, d), ("pageable D2H", d, hq)):
        best = 1e9
        for _ in range(8):
            torch.cuda.synchronize(); t = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        print(f"{mb} MB {name}: {mb / 1024 / best:.1f} GB/s ({1e3 * best:.2f} ms)")
import numpy as np
a = np.empty(24 * (1 << 20) // 8); b = np.empty_like(a)
best = 1e9
for _ in range(8):
    t = time.perf_counter(); b[:] = a; best = min(best, time.perf_counter() - t)
print(f"24 MB host memcpy (one thread): {24 / 1024 / best:.1f} GB/s")
