import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import psf_training, synth_data
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
for problem in ("order",):
    net = psf_training.build_model(problem, 16384).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    loss = torch.nn.CrossEntropyLoss() if problem == "order" else torch.nn.MSELoss()
    X, Y = psf_training.make_split(problem, 40, 16384, dev, 1)
    def step():
        opt.zero_grad(set_to_none=True)
        out = loss(net(X).squeeze(), Y); out.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); print(problem, "ms/step", (time.perf_counter() - t0) / 5 * 1e3)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3): step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
