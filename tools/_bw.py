import torch, time
dev=torch.device("cuda:0")
n=1<<30  # 1 Gi halves = 2 GiB
a=torch.empty(n,dtype=torch.float16,device=dev); b=torch.empty(n,dtype=torch.float16,device=dev)
def t(fn,reps=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ms=t(lambda: a.fill_(1.0)); print("fill 2 GiB: %.3f ms  %.2f TB/s (write)"%(ms, 2*n/ms/1e9))
ms=t(lambda: b.copy_(a)); print("copy 2 GiB: %.3f ms  %.2f TB/s (read+write)"%(ms, 4*n/ms/1e9))
ms=t(lambda: a.sum()); print("sum 2 GiB: %.3f ms  %.2f TB/s (read)"%(ms, 2*n/ms/1e9))
c=torch.empty(3,n//4,dtype=torch.float16,device=dev); s=a[:n//4]
ms=t(lambda: c.copy_(s.expand(3,-1))); print("1 read -> 3 writes of 0.5 GiB: %.3f ms  %.2f TB/s total"%(ms, 4*(n//4)*2/ms/1e9))
