import torch
d=torch.device("cuda:0")
x=torch.empty(256*3*416*416,device=d)
y=torch.empty(256*3*416*416,device=d)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True);e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record();torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
b=x.numel()*4
ms=t(lambda:x.zero_()); print("zero_ %.4f ms %.0f GB/s"%(ms,b/ms/1e6))
ms=t(lambda:x.fill_(1.5)); print("fill_ %.4f ms %.0f GB/s"%(ms,b/ms/1e6))
ms=t(lambda:y.copy_(x)); print("copy %.4f ms %.0f GB/s"%(ms,2*b/ms/1e6))
ms=t(lambda:x.sum()); print("sum(read) %.4f ms %.0f GB/s"%(ms,b/ms/1e6))
