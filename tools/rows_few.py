"""Row softmax with few, long rows (e.g. logits over a vocabulary): time per call against torch."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flash-attention_amd"))
import my_flash_attn_cuda as m

def med(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

for h, w in ((1, 32768), (16, 32768), (16, 131072), (64, 131072), (1, 1 << 20), (8, 262144), (4, 1 << 22), (128, 32768), (256, 50257), (32, 151936)):
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(h, w, device="cuda").to(dt)
        t = med(lambda: m.softmax(x, dim=-1))
        t2 = med(lambda: torch.softmax(x, dim=-1))
        err = (m.softmax(x, dim=-1).float() - torch.softmax(x.float(), dim=-1)).abs().max().item()
        nb = 2 * x.numel() * x.element_size()
        print(f"[{h},{w}] {str(dt)[6:]:8s}: {t*1e3:8.1f} us {nb/t/1e9:6.2f} TB/s | torch {t2*1e3:8.1f} us | maxerr {err:.1e}", flush=True)
