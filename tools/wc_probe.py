import sys, time
sys.path.insert(0, ".")
import torch
import allegro_amd.ops
dev = torch.device("cuda:0")
E, u, l = 298144, 64, 2
sh = torch.randn(E, 9, device=dev); w = torch.randn(E, u * 3, device=dev); t = torch.randn(E, u, 9, device=dev)
def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for which, a, b in ((0, sh, w), (1, t, sh), (2, t, w)):
    print("weighted_channels which", which, round(tm(lambda: torch.ops.allegro_amd.weighted_channels(which, a, b, u, l, False, 0)), 1), "us")
