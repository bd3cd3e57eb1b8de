"""Which rfft shapes make rocFFT compile kernels at run time (first call in a fresh process)?  Dev probe.
usage: fft_jit_probe.py <variant>   a: n=10000 along dim 1 of [1, 5000, 12000] (what diagnostics.py did in round 2)
                                    b: n=16384 along the LAST dim of a contiguous [12000, 5000]
                                    c: n=16384 along dim 1 of [1, 5000, 12000]
                                    d: n=10000 along the last dim of [12000, 5000]"""
import sys, time, torch
v = sys.argv[1]
x = torch.randn(1, 5000, 12000, device='cuda')
xt = x[0].t().contiguous()
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    if v == 'a':
        f = torch.fft.rfft(x, n=10000, dim=1); a = torch.fft.irfft(f * f.conj(), n=10000, dim=1)
    elif v == 'b':
        f = torch.fft.rfft(xt, n=16384, dim=1); a = torch.fft.irfft(f * f.conj(), n=16384, dim=1)
    elif v == 'c':
        f = torch.fft.rfft(x, n=16384, dim=1); a = torch.fft.irfft(f * f.conj(), n=16384, dim=1)
    else:
        f = torch.fft.rfft(xt, n=10000, dim=1); a = torch.fft.irfft(f * f.conj(), n=10000, dim=1)
    torch.cuda.synchronize()
    print(v, 'call', rep, '%.3f s' % (time.perf_counter() - t0), flush=True)
