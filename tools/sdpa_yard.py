import torch, time
import torch.nn.functional as F
dev='cuda'
for (B,H,L,D) in ((22,32,1471,128),(16,32,2048,128),(30,32,1087,128)):
    q=torch.randn(B,H,L,D,device=dev,dtype=torch.bfloat16); k=torch.randn_like(q); v=torch.randn_like(q)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    for name,ctx in (("default",None),):
        try:
            for _ in range(3):
                o=F.scaled_dot_product_attention(q,k,v,is_causal=True)
            torch.cuda.synchronize()
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            n=10
            e0.record()
            for _ in range(n): o=F.scaled_dot_product_attention(q,k,v,is_causal=True)
            e1.record(); torch.cuda.synchronize()
            ms=e0.elapsed_time(e1)/n
            fl=4*B*H*L*L*D/2
            do=torch.randn_like(o)
            o.backward(do,retain_graph=True); torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                o.backward(do,retain_graph=True)
            e1.record(); torch.cuda.synchronize()
            msb=e0.elapsed_time(e1)/n
            print(f"SDPA causal B{B} H{H} L{L} D{D}: fwd {ms:.3f} ms {fl/ms/1e9:.0f} TF/s | bwd {msb:.3f} ms {2.5*fl/msb/1e9:.0f} TF/s", flush=True)
        except Exception as e:
            print("failed", repr(e)[:300])
