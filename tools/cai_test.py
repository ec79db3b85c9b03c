import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
pkg = load_package()
f = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=8)
for p in [(100,100),(150,120)]: f.addFeature(p)
mu_ptr, s_ptr, ld = f.device_pointers()
class Wrap:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}
t = torch.as_tensor(Wrap(mu_ptr, (26,), "<f4"), device="cuda")
print(t.device, t.dtype, t[:8].cpu().numpy(), f.getFullState()[:8])
t[0] = 5.0
torch.cuda.synchronize()
print("after write via torch:", f.getFullState()[0])
S = torch.as_tensor(Wrap(s_ptr, (26, ld), "<f4"), device="cuda")
print(S.shape, S[13,13].item(), S.data_ptr() == s_ptr)
