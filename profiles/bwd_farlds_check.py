import sys, torch
sys.path.insert(0, "/root/repo")
import sparsefactorization_amd as sfa
from sparsefactorization_amd import chord
dev = torch.device("cuda:0")
B, N, L, C = 40, 16384, 15, 8
g = torch.Generator(device=dev).manual_seed(0)
W = 0.1 * torch.randn(B, N, L, device=dev, generator=g); V = torch.randn(B, N, C, device=dev, generator=g); dZ = torch.randn(B, N, C, device=dev, generator=g)
outs = {}
for abl in (0, 256):
    sfa.set_tuning("bwd_ablate", abl)
    dW, dV = torch.empty_like(W), torch.empty_like(V)
    chord._launch_bwd(dZ, W, V, dW, dV, B, N, L, C, N * C, None)
    torch.cuda.synchronize()
    outs[abl] = (dW, dV)
sfa.set_tuning("bwd_ablate", 0)
print("far rows through LDS: dV bit-equal", torch.equal(outs[0][1], outs[256][1]), " dW bit-equal", torch.equal(outs[0][0], outs[256][0]))
