"""A few column-softmax shapes, 30 calls each, for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/cols_trace.py"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flash-attention_amd"))
import my_flash_attn_cuda as m
for shape, dt in (((8192, 1024), torch.bfloat16), ((8192, 1024), torch.float32), ((8192, 256), torch.float32)):
    x = torch.randn(*shape, device="cuda").to(dt)
    for _ in range(30):
        m.softmax(x, dim=0)
    torch.cuda.synchronize()
