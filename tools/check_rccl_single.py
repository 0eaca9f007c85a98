import os, sys, torch, torch.distributed as dist
sys.path[:0]=["/root/repo","/root/repo/revisit-bpr_amd"]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29701", RANK="0", WORLD_SIZE="1")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
from revisit_bpr.distributed import ItemSync
Q=torch.randn(20109,128,device=dev); Q0=Q.clone()
s=ItemSync([Q])
Q+=1.0
s.start(); Q+=0.5; s.finish()
torch.cuda.synchronize()
print("async ok", torch.allclose(Q, Q0+1.5))
s.sync(); print("sync ok", torch.allclose(Q, Q0+1.5))
import time
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(50): s.finish(); s.start()
s.finish(); torch.cuda.synchronize(); print("per start/finish cycle: %.1f us"%((time.perf_counter()-t)/50*1e6))
x = torch.ones(1 << 20, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize(); print("rccl all_reduce on a side stream ok", float(x.sum()) == float(1 << 20))
dist.barrier(); dist.destroy_process_group(); print("done")
