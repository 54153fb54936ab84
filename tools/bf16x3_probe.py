import torch, time
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
M, K, N = 65536, 384, 1536
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
ref = (x.double() @ w.double().t())
t32 = timeit(lambda: torch.mm(x, w.t()))
o32 = torch.mm(x, w.t())
print(f"fp32 mm: {t32:.1f} us, err {((o32 - ref).abs().max() / ref.abs().max()).item():.2e}")
def split3(a):
    hi = a.to(torch.bfloat16); r = a - hi.float()
    mid = r.to(torch.bfloat16); r = r - mid.float()
    lo = r.to(torch.bfloat16)
    return hi, mid, lo
try:
    xh, xm, xl = split3(x); wh, wm, wl = split3(w)
    A6 = torch.cat([xh, xh, xm, xh, xl, xm], 1).contiguous(); B6 = torch.cat([wh, wm, wh, wl, wh, wm], 1).contiguous()
    A3 = torch.cat([xh, xh, xm], 1).contiguous(); B3 = torch.cat([wh, wm, wh], 1).contiguous()
    for name, A, B in (("6 products", A6, B6), ("3 products", A3, B3)):
        o = torch.mm(A, B.t(), out_dtype=torch.float32)
        t = timeit(lambda: torch.mm(A, B.t(), out_dtype=torch.float32))
        print(f"bf16 split {name}: K'={A.shape[1]} {t:.1f} us  err {((o - ref).abs().max() / ref.abs().max()).item():.2e}  out {o.dtype}")
    ts = timeit(lambda: split3(x))
    print(f"split3 of x with torch ops: {ts:.1f} us")
except Exception as e:
    print("out_dtype path failed:", repr(e)[:300])
# plain bf16 GEMM rates at large K for reference
for K2 in (384, 1152, 2304):
    a = torch.randn(M, K2, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K2, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: torch.mm(a, b.t()))
    print(f"bf16 mm (bf16 out) M={M} K={K2} N={N}: {t:.1f} us = {2*M*K2*N/t*1e-6:.0f} TF/s")

# fp16 pieces (11-bit mantissa each: two pieces carry 22 bits), per-tensor power-of-two scaling against underflow
def split2h(a):
    s = 2.0 ** torch.floor(torch.log2(1.0 / a.abs().max())).item()
    a = a * s
    hi = a.half(); lo = (a - hi.float()).half()
    return hi, lo, s
xh, xl, sx = split2h(x); wh, wl, sw = split2h(w)
A3 = torch.cat([xh, xh, xl], 1).contiguous(); B3 = torch.cat([wh, wl, wh], 1).contiguous()
o = torch.mm(A3, B3.t(), out_dtype=torch.float32) / (sx * sw)
t = timeit(lambda: torch.mm(A3, B3.t(), out_dtype=torch.float32))
print(f"fp16 split, 2 pieces / 3 products: K'={A3.shape[1]} {t:.1f} us  err {((o - ref).abs().max() / ref.abs().max()).item():.2e}")
ts = timeit(lambda: split2h(x))
print(f"split of x with torch ops: {ts:.1f} us")
