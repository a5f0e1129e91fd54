import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch, my_flash_attn_cuda as mfa
def med(fn, it=20):
    for _ in range(3): fn()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    torch.cuda.synchronize()
    for a,b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
torch.manual_seed(1)
for n,d in [(128,64),(1024,64),(4096,128)]:
    q,k,v=(torch.randn(n,d,device="cuda") for _ in range(3)); kt=k.t().contiguous()
    ms=med(lambda: mfa.my_attention(q,kt,v))
    ms_t=med(lambda: torch.softmax(q@kt,1)@v)
    print(f"my_attention f32 N={n} d={d}: {ms*1e3:8.1f} us  {4*n*n*d/ms/1e9:8.2f} TF   torch naive {ms_t*1e3:8.1f} us")
for B,H,N,d in [(4,32,1024,64),(4,32,4096,128)]:
    q,k,v=(torch.randn(B,H,N,d,device="cuda") for _ in range(3))
    ms=med(lambda: mfa.flash_attn_fwd(q,k,v,scale=d**-0.5),5)
    ms_t=med(lambda: torch.nn.functional.scaled_dot_product_attention(q,k,v),5)
    print(f"flash_attn_fwd f32 {B,H,N,d}: {ms:8.3f} ms  {4*B*H*N*N*d/ms/1e9:8.2f} TF   torch sdpa f32 {ms_t:8.3f} ms")
