import torch, time
dev = torch.device('cuda:0')
def run(nbytes_mb, nstream, reps=20):
    n = nbytes_mb * 1024 * 1024 // 4
    xs = [torch.randn(n, device=dev) for _ in range(nstream)]
    ys = [torch.empty_like(x) for x in xs]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    torch.cuda.synchronize()
    def once():
        for s, x, y in zip(streams, xs, ys):
            with torch.cuda.stream(s):
                torch.add(x, 1.0, out=y)
    for _ in range(3): once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, nstream * 2 * n * 4 / dt / 1e12
for mb in (8, 32, 128):
    a = run(mb, 1); b = run(mb, 2); c = run(mb, 4)
    print('%4d MB tensors: 1 stream %.1f us %.2f TB/s | 2 streams %.1f us %.2f TB/s | 4 streams %.1f us %.2f TB/s' % (mb, a[0]*1e6, a[1], b[0]*1e6, b[1], c[0]*1e6, c[1]))
