"""HBM write / copy rates of plain streaming kernels on this card (what a perfectly overlapped D store could reach)."""
import torch
dev = "cuda"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mb in (64, 256, 541, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty(n, dtype=torch.float32, device=dev)
    ms_fill = t(lambda: a.fill_(1.0))
    ms_copy = t(lambda: b.copy_(a))
    ms_read = t(lambda: a.sum())
    print("%5d MB: fill %.3f ms = %.2f TB/s write | copy %.3f ms = %.2f TB/s r+w | sum %.3f ms = %.2f TB/s read"
          % (mb, ms_fill, mb * 1.048576e-3 / ms_fill, ms_copy, 2 * mb * 1.048576e-3 / ms_copy, ms_read, mb * 1.048576e-3 / ms_read), flush=True)
