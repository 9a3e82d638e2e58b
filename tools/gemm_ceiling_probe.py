"""What does the vendor GEMM (hipBLASLt behind torch.matmul) sustain on THIS box, on random data, for seconds at a time?
A measured ceiling for "executed fp16 MFMA TFLOP/s" to set beside the conv kernel's ~920 TF executed: square shapes and the
GEMM shapes of the network's three heaviest 3x3 layers at bs=64 (M = B*Ho*Wo, N = Cout, K = 9*Cin).  Probe only (torch is
used as a measuring stick, nothing here is on the product path)."""
import sys, time, json
import torch

def run(M, N, K, dtype, fill, secs=2.0):
    a = torch.empty(M, K, device="cuda", dtype=dtype); b = torch.empty(K, N, device="cuda", dtype=dtype)
    if fill == "randn": a.normal_(); b.normal_()
    elif fill == "zeros": a.zero_(); b.zero_()
    c = torch.empty(M, N, device="cuda", dtype=dtype)
    for _ in range(3): torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    # sustained: keep launching for `secs`, time the last half
    n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs / 2:
        for _ in range(10): torch.matmul(a, b, out=c)
        torch.cuda.synchronize(); n += 10
    per = (time.perf_counter() - t0) / n
    iters = max(10, int(secs / 2 / per))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): torch.matmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9

shapes = [("8192^3", 8192, 8192, 8192), ("16384x8192x8192", 16384, 8192, 8192),
          ("128->256 3x3 @52^2 bs64", 64 * 2704, 256, 1152), ("256->512 3x3 @26^2 bs64", 64 * 676, 512, 2304),
          ("512->1024 3x3 @13^2 bs64", 64 * 169, 1024, 4608), ("64->128 3x3 @104^2 bs64", 64 * 10816, 128, 576)]
out = []
for name, M, N, K in shapes:
    for dtype in (torch.float16, torch.bfloat16):
        for fill in ("randn", "zeros"):
            tf = run(M, N, K, dtype, fill)
            rec = dict(shape=name, M=M, N=N, K=K, dtype=str(dtype).split(".")[1], fill=fill, tflops=round(tf, 1))
            print(json.dumps(rec), flush=True); out.append(rec)
