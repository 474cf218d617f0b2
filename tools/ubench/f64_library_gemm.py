"""What the vendor library's fp64 GEMM sustains on the C5 linear-layer shapes (M = 1.72e6 edge rows): the yardstick for
gemm_f64_rows_kernel / the staged fp64 kernel (aa_gemm.hip).  python tools/ubench/f64_library_gemm.py"""
import torch

M = 1_720_000
dev = torch.device("cuda:0")
for K, N in [(128, 128), (256, 128), (384, 128), (512, 128), (640, 128), (128, 640)]:
    a = torch.randn(M, K, dtype=torch.float64, device=dev)
    w = torch.randn(K, N, dtype=torch.float64, device=dev)
    out = torch.empty(M, N, dtype=torch.float64, device=dev)
    for _ in range(3):
        torch.mm(a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.mm(a, w, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"torch.mm f64 {K:4d} x {N:4d}: {ms:7.3f} ms  {2.0 * M * K * N / ms * 1e-9:6.1f} TFLOP/s  {(M * (K + N) * 8) / ms * 1e-9:6.2f} TB/s", flush=True)
    del a, w, out
