import torch,time
dev=torch.device("cuda:0")
for mb in (200, 500, 1200):
    n=mb*1024*1024//4
    xs=[torch.randn(n,device=dev) for _ in range(4)]
    ys=[torch.empty_like(xs[0]) for _ in range(4)]
    hs=[torch.empty(n,dtype=torch.float16,device=dev) for _ in range(4)]
    for name,fn,bytes_ in (("copy fp32->fp32", lambda i: ys[i%4].copy_(xs[i%4]), 8*n), ("cast fp32->fp16", lambda i: hs[i%4].copy_(xs[i%4]), 6*n), ("read-only sum", lambda i: xs[i%4].sum(), 4*n), ("fill", lambda i: ys[i%4].fill_(1.0), 4*n)):
        for i in range(8): fn(i)
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): fn(i)
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)/40*1e3
        print(f"{mb} MB {name}: {us:.1f} us {bytes_/us/1e6:.2f} TB/s")
