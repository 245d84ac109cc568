import torch, time
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (128, 512, 2048):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    tf = t(lambda: a.zero_()); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    print(f"{mb} MB: fill {mb / 1024 / tf / 1e3 * 1.0737:.2f} TB/s  copy (r+w) {2 * mb / 1024 / tc / 1e3 * 1.0737:.2f} TB/s  read(sum) {mb / 1024 / tr / 1e3 * 1.0737:.2f} TB/s")
