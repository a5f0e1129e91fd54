"""16-bit column softmax with the 1024-thread register-resident kernel on or off (tuning flavour)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
from my_flash_attn_cuda import _lib
t = Tuned()
TAG = {torch.float16: _lib.MYFA_F16, torch.bfloat16: _lib.MYFA_BF16}
def call(x, out, outer, ln, inner, ws):
    nb = t.lib.myfa_softmax_nd_ws_bytes(outer, ln, inner, TAG[x.dtype])
    if nb > 0 and (ws is None or ws.numel() < nb):
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    t._check(t.lib.myfa_softmax_nd_ws(x.data_ptr(), out.data_ptr(), outer, ln, inner, TAG[x.dtype], 1.0, None, 1, 0,
                                      ws.data_ptr() if nb else None, nb, torch.cuda.current_stream().cuda_stream))
    return ws
for shape, dim in (((32, 1024, 4096), 1), ((1024, 16384), 0), ((16, 16, 1024, 1024), 2), ((64, 512, 8192), 1), ((8, 8, 1000, 1024), 2)):
    outer = 1
    for n in shape[:dim]: outer *= n
    inner = 1
    for n in shape[dim + 1:]: inner *= n
    ln = shape[dim]
    xs = [torch.randn(*shape, device="cuda").to(torch.bfloat16) for _ in range(3)]
    out = torch.empty_like(xs[0])
    ref = torch.softmax(xs[0].double(), dim)
    line = f"{shape} dim {dim} bf16:"
    for knob in (-1, 0):
        t.set(softmax_cols_big=knob)
        ws = call(xs[0], out, outer, ln, inner, None)
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().max().item()
        res = []
        for rnd in range(3):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for i in range(3): ws = call(xs[i % 3], out, outer, ln, inner, ws)
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(ev):
                a.record(); ws = call(xs[i % 3], out, outer, ln, inner, ws); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev); res.append(ts[len(ts) // 2])
        ms = sorted(res)[1]
        line += f"  big {knob:2d}: {ms*1e3:7.1f} us {2*xs[0].numel()*2/ms/1e9:5.2f} TB/s (err {err:.0e})"
    print(line, flush=True)
