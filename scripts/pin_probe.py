import time, torch
torch.cuda.init()
n=110080
xs_p=[torch.randn(1,n).pin_memory() for _ in range(400)]
xs_n=[torch.randn(1,n) for _ in range(400)]
for name,xs in (("pinned",xs_p),("pageable",xs_n)):
    t=time.perf_counter(); ok=all(bool(torch.isfinite(x).all()) for x in xs); dt=time.perf_counter()-t
    print(name, "isfinite pass over", 400*n*4/1e6, "MB:", round(dt*1e3,1), "ms")
    t=time.perf_counter(); s=sum(float(x.abs().max()) for x in xs); dt=time.perf_counter()-t
    print(name, "abs-max pass:", round(dt*1e3,1), "ms")
