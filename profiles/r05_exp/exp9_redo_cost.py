import torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import flashattention_c_amd as fa
dev = torch.device('cuda:0')
def t(q,k,v,causal=False,**kw):
    return fa.time_forward(q,k,v,causal,warmup=5,iters=10,**kw)
for (bh,n,d) in ((128,1024,64),(64,2048,64),(32,4096,64),(16,8192,64),(128,1024,128),(128,1024,32)):
    q,k,v=(torch.randn(bh,n,d,device=dev) for _ in range(3))
    tiny=v*2.0**-60
    a=fa.stats()['tiles_redone']
    base=t(q,k,v); slow=t(q,k,tiny)
    torch.cuda.synchronize(); b=fa.stats()['tiles_redone']
    row=f"fp32 {bh}x{n}x{d}: N(0,1) {base:.4f} ms, tiny V {slow:.4f} ms ({slow/base:.1f}x), tiles redone per forward {(b-a)//15}"
    for kern in ("split:1","split:3","split:4"):
        try:
            row+=f" | {kern}: {t(q,k,v,kernel=kern):.4f} / {t(q,k,tiny,kernel=kern):.4f}"
        except Exception as e:
            row+=f" | {kern}: n/a"
    print(row, flush=True)
    qb,kb,vb=(x.to(torch.bfloat16) for x in (q,k,v)); tb=(vb.float()*2.0**-60).to(torch.bfloat16)
    print(f"   bf16: N(0,1) {t(qb,kb,vb):.4f} ms, tiny V {t(qb,kb,tb):.4f} ms")
