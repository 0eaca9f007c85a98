import sys, time, cProfile, pstats
sys.path[:0]=["/root/repo","/root/repo/revisit-bpr_amd"]
import torch
from revisit_bpr.models import BPR
from revisit_bpr.models.bpr import MF
dev=torch.device("cuda")
U,I,d,B=9950,4826,64,256
torch.manual_seed(1)
m=BPR(fuse_forward=True, reg_alphas={"user":0.0016,"item":0.0001,"neg":0.00375}, logits_model=MF(torch.nn.Embedding(U,d,padding_idx=0), torch.nn.Embedding(I,d,padding_idx=0))).to(dev)
opt=torch.optim.SGD(m.parameters(), lr=0.05); m.train()
n=B*300
users=torch.randint(1,U,(n,),device=dev); items=torch.randint(1,I,(n,1),device=dev); neg=torch.randint(1,I,(n,1),device=dev)
def loop():
    for lo in range(0,n,B):
        out=m({"user":users[lo:lo+B],"item":items[lo:lo+B],"neg":neg[lo:lo+B]})
        out["loss"].backward(); opt.step(); opt.zero_grad()
loop(); torch.cuda.synchronize()
t=time.perf_counter(); loop(); torch.cuda.synchronize(); print("us/batch", (time.perf_counter()-t)/300*1e6)
pr=cProfile.Profile(); pr.enable(); loop(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
