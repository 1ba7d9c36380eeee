import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib as L
lib = L.lib()
B, H, W, C = 64, 128, 64, 8
t = torch.randn(B, H, W, C, device="cuda").bfloat16()
out = torch.zeros(C, device="cuda")
def run(): L.check(lib.sx_channel_sum(1, B, H, W, H, W, C, C, L.ptr(t), C, L.ptr(out), L.stream_ptr()))
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); e1.synchronize()
print("channel_sum us:", e0.elapsed_time(e1) * 1e3 / 50, float((out / 55 - t.float().sum((0, 1, 2))).abs().max()))
