"""Diagnostic (library built with PSF_HIPCC_EXTRA=-DPSF_MIXER_DEBUG): the step kernel compares every LDS window entry with
its global source right after the first barrier (counters 0..) and again before the accumulate phase (counters 128..)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mixer, fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [("genome_nores", 16, 16384, 32, 32, 32, 15, False)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, B, N, E, h, C, L, res in SHAPES:
    torch.manual_seed(0)
    g = MLPBlock([h, 'GELU'], E, C).to(dev)
    f = MLPBlock([h, 'GELU'], E, L).to(dev)
    x = torch.randn(B, N, E, device=dev)
    bad = 0
    tot = torch.zeros(1024, dtype=torch.int64)
    with torch.no_grad():
        outs = fused_mlp.fused_mlp_forward(x, [g, f])
        ref = sfa.chord_chain([outs[1]], outs[0], res)
        for rep in range(reps):
            keep = []
            got = fused_mixer.mixer_forward(x, g, [f], res, debug_ws=keep)
            cnt = keep[0][-1024:].view(torch.int32).cpu().to(torch.int64)
            tot += cnt
            if not torch.equal(ref, got):
                bad += 1
                print(f"  {name} rep {rep}: output differs; probe after barrier 1: entries {int(cnt[0])} dwords {cnt[1:5].tolist()} "
                      f"| probe before phase 2: entries {int(cnt[128])} dwords {cnt[129:133].tolist()} | far regs stale {int(cnt[256])} | second pass disagrees {int(cnt[600])} dwords {cnt[601:605].tolist()} lanes {[i for i in range(64) if cnt[608 + i]]} (j,wave) {cnt[672:680].tolist()} "
                      f"first {keep[0][-1024:][700].item():+.5f} second {keep[0][-1024:][701].item():+.5f}", flush=True)
    print(f"{name}: {bad} of {reps} runs differ", flush=True)
    print(f"  far registers before phase 2: stale {int(tot[256])}, by dword {tot[257:261].tolist()}, by lane {tot[264:328].tolist()}, "
          f"by (j, f) {tot[328:344].tolist()}, by wave {tot[356:360].tolist()}", flush=True)
    print(f"  second pass over the sums disagrees: {int(tot[600])}, by dword {tot[601:605].tolist()}, by lane {tot[608:672].tolist()}, "
          f"by (j, wave) {tot[672:680].tolist()}, window entry wrong at re-read by k {tot[690:700].tolist()}", flush=True)
    for base, tag in ((0, "after barrier 1"), (128, "before phase 2")):
        print(f"  {tag}: stale entries {int(tot[base])}, by dword {tot[base + 1:base + 5].tolist()}, by lane {tot[base + 8:base + 72].tolist()}, "
              f"by (pass, wave) {tot[base + 72:base + 72 + 16].tolist()}", flush=True)
