import torch
B,k=32768,35
xs=[torch.randn(B,k,k,device="cuda") for _ in range(3)]
y=torch.empty_like(xs[0])
def timed(fn,n=30):
    for j in range(3): fn(j)
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for j in range(n): fn(j)
    b.record(); b.synchronize()
    return a.elapsed_time(b)/n*1e3
print("copy 160MB->160MB", timed(lambda j: y.copy_(xs[j%3])))
print("sum 160MB", timed(lambda j: xs[j%3].sum()))
print("fill 160MB", timed(lambda j: y.fill_(1.0)))
z=torch.empty(B*k*k//4*4, device="cuda")
print("mul_ 160MB rw", timed(lambda j: xs[j%3].mul_(1.0001)))
