"""Probe for DESIGN §9 item 6(b): an fp32 lm_head evaluated as 2-3 bf16 MFMA GEMMs with fp32 accumulation.
hidden states are bf16 (exact); the fp32 weight is split W = W_hi + W_lo (+ W_lo2) into bf16 terms.
Reports accuracy against the fp32 GEMM (and against an fp64 reference on a slice) and the time of each form."""
import torch

dev = torch.device("cuda", 0)
T, H, V = 8192, 3584, 152064
torch.manual_seed(0)
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
w = (torch.randn(V, H, device=dev) * 0.02)


def split(w32, terms):
    out, r = [], w32.clone()
    for _ in range(terms):
        p = r.to(torch.bfloat16)
        out.append(p)
        r = r - p.float()
    return out


def timeit(fn, iters=3):
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


out = torch.empty(T, V, device=dev, dtype=torch.float32)
xf = x.float()
ms32 = timeit(lambda: torch.mm(xf, w.t(), out=out))
ref = out.clone()
sl = slice(0, 256)
ref64 = (xf[sl].double() @ w.double().t())
print(f"fp32 GEMM: {ms32:.2f} ms; max |fp32 - fp64| on 256 rows = {(ref[sl].double() - ref64).abs().max().item():.3e}")
for terms in (1, 2, 3):
    parts = split(w, terms)

    def run():
        acc = torch.mm(x, parts[0].t(), out_dtype=torch.float32)
        for p in parts[1:]:
            acc += torch.mm(x, p.t(), out_dtype=torch.float32)
        return acc

    try:
        got = run()
    except Exception as e:  # noqa: BLE001
        print(f"{terms}-term split: not supported here: {type(e).__name__}: {e}")
        break
    ms = timeit(run)
    err32 = (got - ref).abs().max().item()
    err64 = (got[sl].double() - ref64).abs().max().item()
    print(f"{terms}-term bf16 split: {ms:.2f} ms ({ms32 / ms:.2f}x the fp32 GEMM rate); max |split - fp32| = {err32:.3e}; "
          f"max |split - fp64| on 256 rows = {err64:.3e}; logits scale = {ref.abs().max().item():.2f}")
    del got, parts
    torch.cuda.empty_cache()
