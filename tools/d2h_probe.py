import torch, time
n=3840*2160*3
d=torch.zeros(n,dtype=torch.uint8,device='cuda')
h=torch.empty(n,dtype=torch.uint8).pin_memory()
s=torch.cuda.Stream()
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(20): h.copy_(d,non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print(f"D2H {n/1e6:.1f} MB: {dt*1e3:.3f} ms  {n/dt/1e9:.1f} GB/s")
