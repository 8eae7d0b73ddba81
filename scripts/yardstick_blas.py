"""Dev yardstick (NOT product): what the vendor BLAS reaches on the path's GEMM shapes."""
import torch
M = 24576
for (m, n, k, name) in [(M, 1152, 384, "qkv"), (M, 384, 384, "proj"), (M, 1536, 384, "fc1"), (M, 384, 1536, "fc2"), (8192, 8192, 8192, "big")]:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.nn.functional.linear(a, w, bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): torch.nn.functional.linear(a, w, bias)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"{name:5s} {ms*1e3:8.1f} us {2*m*n*k/ms/1e9:8.1f} TFLOP/s")
