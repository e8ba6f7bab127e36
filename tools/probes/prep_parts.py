import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.point_prep_bench import gpu_ms
n = 300000
d = torch.rand(n, 5, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
m = d[:, 0] > 0.2
print("randperm", gpu_ms(lambda: torch.randperm(n, device="cuda"), 50))
print("item", gpu_ms(lambda: cnt.item(), 50))
print("nonzero-index", gpu_ms(lambda: d[m], 50))
print("mask", gpu_ms(lambda: (d[:, 0] >= 0.1) & (d[:, 0] <= 0.9) & (d[:, 1] >= 0.1) & (d[:, 1] <= 0.9), 50))
p = torch.randperm(n, device="cuda")
print("gather", gpu_ms(lambda: d[p], 50))
pin = torch.zeros(1, dtype=torch.int32).pin_memory()
def pinned():
    pin.copy_(cnt, non_blocking=True); torch.cuda.current_stream().synchronize(); return int(pin[0])
print("pinned read", gpu_ms(pinned, 50))
