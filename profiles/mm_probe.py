import torch, time
dev="cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
M=65536
for (N,K) in [(432,432),(1024,432),(512,1024),(256,512)]:
    x=torch.randn(M,K,device=dev,dtype=torch.bfloat16); w=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    dy=torch.randn(M,N,device=dev,dtype=torch.bfloat16)
    f=t(lambda: x@w.t()); bx=t(lambda: dy@w); bw=t(lambda: dy.t()@x)
    fl=2*M*N*K
    print(f"N={N} K={K}: fwd {f:.1f}us ({fl/f/1e6:.0f} TF) gradX {bx:.1f}us ({fl/bx/1e6:.0f} TF) gradW {bw:.1f}us ({fl/bw/1e6:.0f} TF)")
    x32=x.float(); w32=w.float()
    f32=t(lambda: x32@w32.t())
    print(f"   fp32 fwd {f32:.1f}us ({fl/f32/1e6:.0f} TF)")
