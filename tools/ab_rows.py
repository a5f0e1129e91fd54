"""A/B of two builds of the library on long-row softmax (the streamed kernel), interleaved:
python tools/ab_rows.py libA.so libB.so"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
from my_flash_attn_cuda import _lib
libs = [Tuned(p) for p in sys.argv[1:]]
TAG = {torch.float32: _lib.MYFA_F32, torch.bfloat16: _lib.MYFA_BF16}
def call(t, x, out, ws):
    h, w = x.shape
    nb = t.lib.myfa_softmax_nd_ws_bytes(h, w, 1, TAG[x.dtype])
    if nb > 0 and (ws is None or ws.numel() < nb):
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    t._check(t.lib.myfa_softmax_nd_ws(x.data_ptr(), out.data_ptr(), h, w, 1, TAG[x.dtype], 1.0, None, 1, 0,
                                      ws.data_ptr() if nb else None, nb, torch.cuda.current_stream().cuda_stream))
    return ws
for (h, w, dt) in ((1024, 131072, torch.float32), (1024, 151936, torch.float32), (4096, 70001, torch.float32), (2048, 262144, torch.bfloat16), (512, 131073, torch.bfloat16), (4, 4194304, torch.float32)):
    xs = [torch.randn(h, w, device="cuda").to(dt) for _ in range(2)]
    out = torch.empty_like(xs[0])
    ref = torch.softmax(xs[0][:4].double(), -1)
    line = f"[{h},{w}] {str(dt)[6:]}:"
    for t, p in zip(libs, sys.argv[1:]):
        ws = call(t, xs[0], out, None)
        torch.cuda.synchronize()
        err = (out[:4].double() - ref).abs().max().item()
        res = []
        for rnd in range(3):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            for i in range(2): ws = call(t, xs[i % 2], out, ws)
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(ev):
                a.record(); ws = call(t, xs[i % 2], out, ws); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev); res.append(ts[len(ts) // 2])
        ms = sorted(res)[1]
        line += f"  {os.path.basename(p)}: {ms*1e3:8.1f} us {2*xs[0].numel()*xs[0].element_size()/ms/1e9:5.2f} TB/s (err {err:.0e})"
    print(line, flush=True)
    del xs, out
    torch.cuda.empty_cache()
