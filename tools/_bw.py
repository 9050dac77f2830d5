import torch, time
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nf in (3276800, 32768000, 268435456):
    a = torch.randn(nf, device="cuda"); b = torch.randn(nf, device="cuda"); c = torch.empty_like(a)
    us = t(lambda: torch.sub(a, b, out=c))
    print("sub   n=%d  %.1f us  %.2f TB/s" % (nf, us, 3 * nf * 4 / us / 1e6))
    us = t(lambda: c.copy_(a))
    print("copy  n=%d  %.1f us  %.2f TB/s" % (nf, us, 2 * nf * 4 / us / 1e6))
# cold-ish: cycle over many buffers so that nothing is cache resident
bufs = [torch.randn(3276800, device="cuda") for _ in range(64)]
out = torch.empty_like(bufs[0])
i = [0]
def f():
    i[0] = (i[0] + 2) % 64
    torch.sub(bufs[i[0]], bufs[i[0] + 1], out=out)
us = t(f, 200)
print("sub cycling 64 x 13MB buffers: %.1f us  %.2f TB/s" % (us, 3 * 3276800 * 4 / us / 1e6))
