import torch, time
torch.cuda.init()
for c in (100000, 1000000):
    torch.cuda._sleep(c); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record(); torch.cuda._sleep(c); b.record(); b.synchronize()
    print(c, "cycles =", a.elapsed_time(b)*1e3, "us")
