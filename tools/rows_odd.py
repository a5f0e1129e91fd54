"""Row softmax at widths that are not a multiple of the 16-byte item (rows not 16-byte aligned):
time per call and algorithmic TB/s beside the next aligned width and torch."""
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

for h, w in ((8192, 1023), (4096, 4097), (8192, 8191), (2048, 30001), (256, 50257), (1024, 50257), (1024, 50264), (4096, 50257), (2048, 65536), (2048, 65535), (4096, 4096), (8192, 8192), (16384, 16384)):
    for dt in (torch.float32, torch.bfloat16):
        xs = [torch.randn(h, w, device="cuda").to(dt) for _ in range(3)]
        i = [0]
        def run():
            i[0] += 1
            return m.softmax(xs[i[0] % 3], dim=-1)
        t = med(run)
        t2 = med(lambda: torch.softmax(xs[0], dim=-1))
        err = (m.softmax(xs[0], dim=-1).float() - torch.softmax(xs[0].float(), dim=-1)).abs().max().item()
        nb = 2 * xs[0].numel() * xs[0].element_size()
        print(f"[{h},{w}] {str(dt)[6:]:8s}: {t*1e3:8.1f} us {nb/t/1e9:6.2f} TB/s | torch {t2*1e3:8.1f} us | maxerr {err:.1e}", flush=True)
