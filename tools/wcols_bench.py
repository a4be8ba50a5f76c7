import sys, torch, ctypes as C
sys.path.insert(0, ".")
from dss_vae_amd import lib as L
lib = L.load()
dev = "cuda:0"
rows, cols, nj = 7936, 30000, 8
slab = torch.rand(rows, cols, device=dev)
scale = torch.rand((cols + 127) // 128, rows, device=dev)
X = torch.rand(rows, 904, device=dev)
Cm = torch.zeros(cols, 904, device=dev)
def run():
    L.check(lib.dssvae_ce_wgrad_cols(L.ptr(slab), cols, rows, cols, L.ptr(scale), rows, C.c_void_p(X.data_ptr() + 4 * 896), 904, nj,
                                     C.c_void_p(Cm.data_ptr() + 4 * 896), 904, 1.0, L.stream()), "x")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("ce_wgrad_cols %.3f ms  %.2f TB/s of slab" % (ms, rows * cols * 4 / ms / 1e9))
