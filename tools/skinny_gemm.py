"""Is a skinny batched SGEMM faster with the long dimension as the BLAS 'm'?  (torch.bmm -> rocBLAS/hipBLASLt)"""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
for (rows, k, n, E) in ((99328, 75, 6, 64), (13968, 150, 16, 64), (97, 576, 120, 256)):
    A = torch.randn(E, rows, k, device='cuda'); B = torch.randn(E, k, n, device='cuda')
    At = A.transpose(1, 2).contiguous(); Bt = B.transpose(1, 2).contiguous()
    def t(f):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5
    t1 = t(lambda: torch.bmm(A, B))            # C [rows, n] row-major  (BLAS m = n: skinny m)
    t2 = t(lambda: torch.bmm(Bt, At))          # C^T [n, rows] row-major (BLAS m = rows)
    t3 = t(lambda: torch.bmm(Bt, A.transpose(1, 2)))   # same product, A given row-major (transposed view)
    fl = 2 * E * rows * k * n
    print(f'rows={rows} k={k} n={n} E={E}: [rows,n] {t1*1e3:7.2f} ms ({fl/t1/1e12:5.2f} TF)  [n,rows] {t2*1e3:7.2f} ms ({fl/t2/1e12:5.2f} TF)  [n,rows] from row-major A {t3*1e3:7.2f} ms ({fl/t3/1e12:5.2f} TF)')
