"""Context for SURVEY §8(f)-1 (fusing the lm_head into the loss): what the library GEMM that produces
the logits costs on this GPU, next to the loss kernel that consumes them.
hidden [T=8192, H=3584] x lm_head [V=152064, H]^T, bf16 and fp32 (the reference keeps the head in fp32)."""
import torch

dev = torch.device("cuda", 0)
T, H, V = 8192, 3584, 152064
flops = 2.0 * T * H * V


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for dt, name in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
    x = torch.randn(T, H, device=dev, dtype=dt)
    w = torch.randn(V, H, device=dev, dtype=dt) * 0.02
    out = torch.empty(T, V, device=dev, dtype=dt)
    ms = timeit(lambda: torch.matmul(x, w.t(), out=out))
    print(f"lm_head forward GEMM {name}: {ms:8.2f} ms  {flops / ms / 1e9:8.1f} TFLOP/s  (writes {out.numel() * out.element_size() / 1e9:.2f} GB of logits)")
    g = torch.randn(T, V, device=dev, dtype=dt)
    ms2 = timeit(lambda: torch.matmul(g, w))           # d hidden
    ms3 = timeit(lambda: torch.matmul(g.t(), x))       # d weight
    print(f"lm_head backward GEMMs {name}: dH {ms2:8.2f} ms  dW {ms3:8.2f} ms")
    del x, w, out, g
    torch.cuda.empty_cache()
