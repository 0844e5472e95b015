import torch, time
dev=torch.device("cuda:0")
for n in (256<<20, 712<<20, 1420<<20):
    a=torch.empty(n, dtype=torch.uint8, device=dev); b=torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): b.copy_(a)
    torch.cuda.synchronize(); dt=(time.time()-t)/10
    print("copy", n>>20, "MiB: %.2f TB/s (read+write)" % (2*n/dt/1e12))
    t=time.time()
    for _ in range(10): a.fill_(1)
    torch.cuda.synchronize(); dt=(time.time()-t)/10
    print("fill", n>>20, "MiB: %.2f TB/s" % (n/dt/1e12))
    x=a.view(torch.float32)
    t=time.time()
    for _ in range(10): s=x.sum()
    torch.cuda.synchronize(); dt=(time.time()-t)/10
    print("read(sum)", n>>20, "MiB: %.2f TB/s" % (n/dt/1e12))
