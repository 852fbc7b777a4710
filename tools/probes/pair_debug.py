import sys, ctypes, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from airpose_amd import _native as Nn
import test_gpu_parity as T
dev = torch.device('cuda', 0)
L = Nn.lib()
n, H, P, N1 = 2, 28, 128, 128
M, C3 = n*H*H, 4*P
t2, x, w3, w1, s3, h3, s1, h1 = T._pair_case(dev, n, H, P, N1, seed=1)
p = lambda t: ctypes.c_void_p(t.data_ptr())
out = torch.zeros(M, C3, dtype=torch.bfloat16, device=dev); t1n = torch.zeros(M, N1, dtype=torch.bfloat16, device=dev)
Nn.check(L.ap_conv_pair_nhwc(p(t2), p(w3), p(s3), p(h3), p(x), p(w1), p(s1), p(h1), p(out), p(t1n), M, P, N1, Nn.stream_ptr(dev)), 'pair')
torch.cuda.synchronize()
acc = t2.double() @ w3.double().T
want = (acc * s3.double() + h3.double() + x.double()).clamp_min(0)
err = (out.double() - want).abs().cpu().numpy()
bad = err > 0.02 * (1 + want.abs().cpu().numpy())
print('bad frac', bad.mean())
print('bad by channel%128 (first 64):', bad.reshape(M, 4, 128).mean((0,1))[:64].round(2))
print('bad by chunk:', bad.reshape(M, 4, 128).mean((0,2)).round(2))
print('bad by pixel%64:', bad.reshape(-1, 64, C3)[:24].mean((0,2)).round(2)) if M % 64 == 0 else print('bad by pixel (first 64):', bad[:64].mean(1).round(2))
# test variants: without residual / without scale
w_nores = (acc * s3.double() + h3.double()).clamp_min(0)
print('match no-res?', ((out.double() - w_nores).abs() < 0.02*(1+w_nores.abs())).double().mean().item())
# partial K: only first 64 / last 64 of K
for nm, sl in (('k0', slice(0,64)), ('k1', slice(64,128))):
    a = t2[:, sl].double() @ w3[:, sl].double().T
    w_ = (a * s3.double() + h3.double() + x.double()).clamp_min(0)
    print(nm, ((out.double() - w_).abs() < 0.02*(1+w_.abs())).double().mean().item())
