import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import synth
from pytextgcn_amd.plan import GraphPlan
cuda = torch.device("cuda:0")
T = [time.perf_counter()]
def lap(name):
    torch.cuda.synchronize(); T.append(time.perf_counter()); print(f"{name}: {T[-1]-T[-2]:.2f} s", flush=True)
N, E, F = 2_000_000, 50_000_000, 200
g = synth.word_doc_graph(N, E, seed=44, device=cuda, features="none"); lap("graph")
plan = GraphPlan(g.edge_index, g.edge_attr, N); lap("plan")
plan2 = GraphPlan(g.edge_index, g.edge_attr, N, degree_sum="reference"); lap("plan reference mode")
del plan2
gen = torch.Generator(device=cuda).manual_seed(1)
x = torch.randn(N, F, device=cuda, generator=gen)
y = torch.randn(N, F, device=cuda, generator=gen)
mx, my = plan.spmm(x), plan.spmm(y); lap("2 spmm")
lin = plan.spmm(2 * x - 3 * y); lap("lin")
lhs = (mx.double() * y.double()).sum().item()
rhs = (x.double() * plan.spmm(y, transpose=True).double()).sum().item(); lap("adjoint")
rp, col, val = plan.export_csr(); lap("export")
rowsum = torch.zeros(N, device=cuda, dtype=torch.float64)
rows = torch.repeat_interleave(torch.arange(N, device=cuda), (rp[1:] - rp[:-1]).long()); lap("repeat_interleave")
rowsum.index_add_(0, rows, val.double()); lap("index_add f64")
ones = plan.spmm(torch.ones(N, 4, device=cuda)); lap("ones spmm")
deg = (rp[1:] - rp[:-1]).long()
sample = torch.cat([deg.topk(8).indices, torch.randint(0, N, (200,), device=cuda, generator=gen)])
for r in sample.tolist():
    s, e = rp[r].item(), rp[r + 1].item()
    ref = (val[s:e].double().unsqueeze(1) * x[col[s:e].long()].double()).sum(0)
lap("sample rows")
