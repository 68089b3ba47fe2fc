import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madeleine_amd import functional as MF
from tests._util import t
dev = torch.device("cuda:0")
M, N, K = 48, 256, 512
a = t((M, K), "bin:a") + 1.5 * torch.sign(t((M, K), "bin:a"))     # magnitudes in [1.5, 2.5]
a = a * (2.0 ** -torch.arange(M, dtype=torch.float32)).unsqueeze(1)
b = 0.05 * t((N, K), "bin:b")
ref = a.double() @ b.double().t()
A, B = MF.split_image(a.to(dev)), MF.split_image(b.to(dev))
C = MF.split_gemm_nt(A, B).double().cpu()
raw = A.data.cpu().contiguous().view(torch.int16).view(M, K // 32, 2, 32).view(torch.float16).double()
dec = (raw[:, :, 0] + raw[:, :, 1]).reshape(M, K) / float(A.scale[0])
for r in range(0, M, 2):
    e = float((C[r] - ref[r]).abs().max() / ref[r].abs().max())
    ei = float((dec[r] - a[r].double()).abs().max() / a[r].abs().max())
    print(f"row 2^-{r:2d}: product rel err {e:.2e}  image rel err {ei:.2e}   (2^(k-39) = {2.0 ** (r - 39):.2e})")
