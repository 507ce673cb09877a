import torch
dev=torch.device("cuda",0)
for mb in (532, 1064):
    x=torch.empty(mb*1000*1000//4, device=dev)
    for name,f in (("fill_",lambda: x.fill_(1.0)),("zero_",lambda: x.zero_())):
        for _ in range(3): f()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/20
        print(mb,name,round(ms,4),"ms",round(mb/1e3/ms,2),"TB/s")
