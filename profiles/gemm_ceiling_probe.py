"""What a tuned library GEMM (torch.matmul = hipBLASLt / rocBLAS) takes on the decoder's GEMM shapes at 1024 rows — a measuring stick for gemm_tile_kernel
(profiles only: the product never calls a library GEMM).  y[R][N] = x[R][K] @ W[N][K]^T, fp16 in, fp32 accumulate, fp16 out."""
import torch, time
shapes = [("qkv", 1024, 3072, 1024), ("o / cross-q / cross-out", 1024, 1024, 1024), ("fc1", 1024, 4096, 1024), ("fc2", 1024, 1024, 4096), ("heads", 1024, 9 * 1088, 1024),
          ("fc1 at 2048 rows", 2048, 4096, 1024), ("o at 2048 rows", 2048, 1024, 1024)]
dev = torch.device("cuda:0")
for name, R, N, K in shapes:
    NB = 24
    x = [torch.randn(R, K, device=dev, dtype=torch.float16) for _ in range(NB)]
    w = [torch.randn(N, K, device=dev, dtype=torch.float16) for _ in range(NB)]
    for i in range(5): y = x[i % NB] @ w[i % NB].t()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()       # replayed as one graph: the eager dispatch of torch (~20 us per call) is not what is measured
    ys = [None] * 96
    with torch.cuda.graph(g):
        for i in range(96): ys[i] = x[i % NB] @ w[i % NB].t()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 96 * 1e3
    print(f"{name:26s} R {R} N {N} K {K}: {us:7.2f} us  {2.0 * R * N * K / us / 1e6:7.1f} TFLOP/s  ({2.0 * R * N * K / us / 1e6 / 2516.8:.3f} of the fp16 peak)")
