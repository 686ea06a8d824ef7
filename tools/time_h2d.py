"""Host -> device upload of one 256 x 256 float64 image as float32: torch's converting .to() against a NumPy cast + plain copy."""
import time, numpy as np, torch
a = np.random.rand(65536)
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
def t(f, n=20):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    return 'min %.0f  median %.0f  max %.0f us' % (min(ts), sorted(ts)[n // 2], max(ts))
print('torch threads', torch.get_num_threads())
print('.to(cuda, float32) from f64      ', t(lambda: torch.from_numpy(a).to(device='cuda', dtype=torch.float32)))
print('numpy cast, then .to(cuda)       ', t(lambda: torch.from_numpy(a.astype(np.float32)).to(device='cuda')))
print('.to(cuda) f64, cast on the device', t(lambda: torch.from_numpy(a).to(device='cuda').to(torch.float32)))
x = torch.rand(65536, device='cuda')
print('.double().cpu().numpy()          ', t(lambda: x.double().cpu().numpy()))
print('.cpu().numpy().astype(f64)       ', t(lambda: x.cpu().numpy().astype(np.float64)))
