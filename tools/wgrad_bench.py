"""`aa_linear_wgrad` (allegro_amd/csrc/aa_train.hip) against the library GEMM for the weight-gradient shapes of the C3 / C4 training
step: x [E,K]^T @ g [E,N], E = edges.  Prints one markdown row per shape: ms of both, the HBM rate of the hand-written kernel
(algorithmic bytes = (K + N) x 4 B per edge), max abs deviation from an fp64 product relative to the largest entry.

    python tools/wgrad_bench.py [E ...] > profiles/rNN_wgrad_bench.md
"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import allegro_amd.ops  # noqa: E402,F401  (registers the ops)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    Es = [int(a) for a in sys.argv[1:]] or [298144, 2725408]
    print("| E | K | N | dtype | aa_linear_wgrad ms | GB/s | torch (rocBLAS / hipBLASLt) ms | speed-up | rel. dev. from fp64 (hand / library) |")
    print("|---|---|---|---|---|---|---|---|---|")
    g = torch.Generator(device=dev).manual_seed(0)
    for E in Es:
        for dtype in (torch.float32, torch.float64):
            for K, N in ((8, 64), (64, 64), (64, 192), (64, 256), (128, 64), (192, 64), (256, 64), (128, 128), (512, 128)):
                if dtype == torch.float64 and E > 10 ** 6 and K * N > 64 * 256:
                    continue
                x = torch.randn(E, K, device=dev, dtype=dtype, generator=g)
                y = torch.randn(E, N, device=dev, dtype=dtype, generator=g)
                t_h = timeit(lambda: torch.ops.allegro_amd.linear_wgrad(x, y, 0))
                t_l = timeit(lambda: x.t() @ y)
                want = x.double().t() @ y.double()
                sc = float(want.abs().max())
                d_h = float((torch.ops.allegro_amd.linear_wgrad(x, y, 0).double() - want).abs().max()) / sc
                d_l = float(((x.t() @ y).double() - want).abs().max()) / sc
                es = 4 if dtype == torch.float32 else 8
                print(f"| {E} | {K} | {N} | {'f32' if es == 4 else 'f64'} | {t_h:.3f} | {E * (K + N) * es / t_h / 1e6:.0f} | {t_l:.3f} | {t_l / t_h:.1f}x | "
                      f"{d_h:.1e} / {d_l:.1e} |", flush=True)
                del x, y, want


if __name__ == "__main__":
    main()
