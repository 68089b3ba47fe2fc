import torch, time
dev = torch.device("cuda:0")
T, H = 262144, 4
E = torch.randn(T, H * 512, device=dev).to(torch.bfloat16)
Wc = torch.randn(H, 1024, 512, device=dev).to(torch.bfloat16) * 0.03
z = torch.empty(T, H, 1024, device=dev, dtype=torch.bfloat16)
dE = torch.zeros(T, H * 512, device=dev, dtype=torch.bfloat16)

def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def fwd_strided():
    for h in range(H):
        torch.mm(E[:, h * 512:(h + 1) * 512], Wc[h].t(), out=z[:, h])
def fwd_bmm():
    return torch.bmm(E.view(T, H, 512).permute(1, 0, 2), Wc.transpose(1, 2))
def dx_strided():
    for h in range(H):
        dE.view(T, H, 512)[:, h].addmm_(z[:, h], Wc[h])
def dw_strided():
    return [torch.mm(z[:, h].t(), E[:, h * 512:(h + 1) * 512]) for h in range(H)]
m0 = torch.cuda.memory_allocated()
print("fwd strided-out mm x4: %.3f ms" % timeit(fwd_strided), "TF", 2*T*H*512*1024/1e9/timeit(fwd_strided))
print("peak extra MB", (torch.cuda.max_memory_allocated() - m0) / 2**20)
zz = fwd_bmm()
print("fwd bmm: %.3f ms" % timeit(fwd_bmm))
ref = torch.stack([E[:, h*512:(h+1)*512].float() @ Wc[h].float().t() for h in range(H)], 1)
fwd_strided()
print("strided result err", float((z.float() - ref).abs().max()), float(ref.abs().max()))
print("dx strided addmm_: %.3f ms" % timeit(dx_strided))
print("dw strided: %.3f ms" % timeit(dw_strided))
x = torch.randn(T, 512, device=dev).to(torch.bfloat16); W1 = torch.randn(512, 512, device=dev).to(torch.bfloat16); W3 = torch.randn(2048, 512, device=dev).to(torch.bfloat16)
b1 = torch.zeros(512, device=dev, dtype=torch.bfloat16); b3 = torch.zeros(2048, device=dev, dtype=torch.bfloat16)
print("linear 512->512: %.3f ms" % timeit(lambda: torch.nn.functional.linear(x, W1, b1)))
print("linear 512->2048: %.3f ms" % timeit(lambda: torch.nn.functional.linear(x, W3, b3)))
try:
    r = torch.mm(z[:, 0].t(), E[:, :512], out_dtype=torch.float32); print("out_dtype ok", r.dtype)
except Exception as ex: print("out_dtype fail", type(ex).__name__, str(ex)[:100])
