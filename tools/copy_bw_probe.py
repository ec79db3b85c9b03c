import torch, time
n=24014; ld=24064
a=torch.empty((n,ld),dtype=torch.float32,device='cuda').normal_()
b=torch.empty_like(a)
for shape in ['full','rows']:
    torch.cuda.synchronize()
    ev0=torch.cuda.Event(enable_timing=True); ev1=torch.cuda.Event(enable_timing=True)
    for it in range(3):
        ev0.record()
        if shape=='full': b.copy_(a)
        else: b[:, :n].copy_(a[:, :n])
        ev1.record(); torch.cuda.synchronize()
        ms=ev0.elapsed_time(ev1)
        byt=2*a.numel()*4 if shape=='full' else 2*n*n*4
        print(shape, f"{ms:.3f} ms  {byt/ms/1e9:.2f} TB/s (read + write)")
