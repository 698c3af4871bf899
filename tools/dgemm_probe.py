import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = 'cuda:0'
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
for (M, N, K) in ((1024, 1024, 32768), (1024, 32768, 1024), (4096, 4096, 4096), (8192, 8192, 8192), (1024, 1024, 262144)):
    A = torch.randn(M, K, dtype=torch.float64, device=dev); B = torch.randn(K, N, dtype=torch.float64, device=dev)
    t = bench(lambda: A @ B)
    print('dgemm %5d x %5d x %6d: %.3f ms  %.1f TF' % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e12))
    Bt = torch.randn(N, K, dtype=torch.float64, device=dev)
    t = bench(lambda: A @ Bt.T)
    print('dgemm NT %5d x %5d x %6d: %.3f ms  %.1f TF' % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e12))
