import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
shdr = importlib.import_module("singlehdr-tf2_amd"); K = shdr._ops
x = torch.randn(16, 512, 512, 4, device="cuda"); w = torch.randn(3, 3, 4, 64, device="cuda") * 0.1; b = torch.randn(64, device="cuda")
w._shdr_const = True
for cout, k in ((64, 3),):
    for _ in range(100): y = K.conv2d(x, w, b, act1=K.ACT_RELU)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): y = K.conv2d(x, w, b, act1=K.ACT_RELU)
    e1.record(); torch.cuda.synchronize()
    print("dbg", os.environ.get("SHDR_X3N_DBG_STORE"), K.conv2d_plan(tuple(x.shape), tuple(w.shape)), "%.4f ms" % (e0.elapsed_time(e1) / 100))
