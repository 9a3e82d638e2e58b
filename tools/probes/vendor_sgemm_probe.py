import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
for (M,N,K) in [(8192,8192,8192),(43264,512,2304),(10816,1024,4608),(173056,256,1152)]:
    a=torch.randn(M,K,device="cuda"); b=torch.randn(K,N,device="cuda"); c=torch.empty(M,N,device="cuda")
    for _ in range(3): torch.matmul(a,b,out=c)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.matmul(a,b,out=c)
    e1.record(); torch.cuda.synchronize()
    print("sgemm %dx%dx%d: %.1f TFLOP/s" % (M,N,K, 2.0*M*N*K/(e0.elapsed_time(e1)/10)/1e9))
