import sys, os, torch
sys.path.insert(0, os.getcwd())
from sparsefactorization_amd.lra_training import train_benchmark
dev = torch.device("cuda:0")
for tag, over in (("reference_config", {}), ("baseline_wording", dict(n_vec=2048, embedding_size=64, n_channels_V=64))):
    for rnd in range(2):
        for graph in (False, True):
            r = train_benchmark("listops", steps=30, warmup=5, device=dev, graph=graph, **over)
            print(tag, "graph" if graph else "eager", f"{r['seconds']*1e3/r['steps']:.3f} ms/step (device {r['event_ms']/r['steps']:.3f})")
