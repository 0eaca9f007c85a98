#!/usr/bin/env python
"""Ten-second RCCL check for a box with >= 2 GPUs (the build container and the 1-GPU boxes never
ran `backend="nccl"` with world > 1):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29650 tools/check_rccl.py

Every rank perturbs its replica of an ML-20M-sized item table with rank-dependent, seeded updates,
runs ItemSync (delta all-reduce on the side stream + fused fold) twice, and rank 0 checks
  * the reconciled table equals Q0 + the sum of every rank's updates (computed locally from the
    seeds, fp32 tolerance: the ring's summation order is RCCL's),
  * the BASE is bit-identical on every rank (the invariant that keeps replicas from drifting),
  * the all-reduce time next to a k_stream step (does it hide?).
Also runs with one rank (world 1: plain copy) and with BPR_DIST_BACKEND=gloo for comparison.
"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from revisit_bpr.distributed import ItemSync  # noqa: E402


def main():
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BPR_DIST_BACKEND", "nccl")
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    I, d = 20109, 128
    g = torch.Generator(device=dev).manual_seed(1)
    Q0 = torch.randn(I, d, device=dev, generator=g) * 0.01
    Q = Q0.clone()
    sync = ItemSync([Q])
    sync.timing = True

    def update(r, k):
        gg = torch.Generator(device=dev).manual_seed(1000 * k + r)
        return torch.randn(I, d, device=dev, generator=gg) * 1e-3

    expect = Q0.double()
    for k in range(2):
        Q += update(rank, k)
        for r in range(world):
            expect += update(r, k).double()
        sync.step()
    sync.finish()
    torch.cuda.synchronize()
    st = sync.timing_read()
    err = (Q.double() - expect).abs().max().item()
    base = sync.base[0]
    same = True
    if world > 1:
        ref = base.clone()
        dist.broadcast(ref, src=0)
        flag = torch.tensor([int(torch.equal(ref, base))], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same = bool(flag.item())
    if rank == 0:
        print(f"backend {backend} world {world}: max |Q - expected| = {err:.3e} (tolerance 1e-6), "
              f"bases bit-identical on every rank: {same}, all-reduce of {st['message_bytes'] / 1e6:.1f} MB: "
              f"{st['all_reduce_ms_avg']} ms avg over {st['all_reduces']} (a k_stream step is ~0.23 ms)")
        assert err < 1e-6 and same
        print("OK")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    t0 = time.time()
    main()
