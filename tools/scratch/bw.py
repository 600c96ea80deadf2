import torch, time
n = 512*752*480
x = torch.randint(0, 1000, (n,), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.sum()); print("sum   %.3f ms  %.2f TB/s read" % (ms, n*4/ms/1e9))
ms = t(lambda: x.max()); print("max   %.3f ms  %.2f TB/s read" % (ms, n*4/ms/1e9))
ms = t(lambda: y.copy_(x)); print("copy  %.3f ms  %.2f TB/s r+w" % (ms, 2*n*4/ms/1e9))
ms = t(lambda: y.fill_(1)); print("fill  %.3f ms  %.2f TB/s write" % (ms, n*4/ms/1e9))
x8 = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
ms = t(lambda: torch.add(x8, 1, out=x8)); 
