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


# ---- the two-tier protocol through the C ABI (bpr_comm_init / bpr_comm_hot_tier / bpr_hot_sync /
# bpr_item_sync over RCCL) against the same protocol with every rank simulated in this process
# (distributed.LocalWorld): the first real node run is this one command.
def two_tier(world, rank, dev):
    import numpy as np

    from revisit_bpr import engine as eng
    from revisit_bpr.distributed import ItemSync, LocalWorld

    U, I, d, n_round, rounds, pieces, H = 2000, 1500, 128, 4096, 3, 2, 64
    rng = np.random.default_rng(5)
    P0 = rng.normal(0, 0.2, (U, d)).astype(np.float32)
    Q0 = rng.normal(0, 0.2, (I, d)).astype(np.float32)
    P0[0] = 0
    Q0[0] = 0
    lens = rng.integers(1, 12, U)
    lens[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(k), replace=False)).astype(np.int32) for k in lens]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    indices = np.concatenate(rows)
    users = np.sort(rng.integers(1, U, world * n_round * rounds)).astype(np.int32)
    pos = (1 + (rng.zipf(1.4, len(users)) % (I - 1))).astype(np.int32)
    counts = torch.bincount(torch.from_numpy(pos).long(), minlength=I)
    cnt = counts.clone()
    cnt[0] = 0
    hot = torch.argsort(cnt * (I + 1) + (I - torch.arange(I)), descending=True)[:H].to(torch.int32)
    u_d, p_d = torch.from_numpy(users).to(dev), torch.from_numpy(pos).to(dev)

    def engine():
        e = eng.Engine(torch.from_numpy(P0).to(dev), torch.from_numpy(Q0).to(dev))
        e.set_reg(0.01, 0.02, 0.03)
        e.set_optimizer(eng.OPT_SGD, lr=0.05)
        e.bind_seen_csr(torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))
        e.set_stream_opts(True, 0)
        return e

    def launch(e, r, k, p):
        lo = (k * world + r) * n_round
        a, b = lo + n_round * p // pieces, lo + n_round * (p + 1) // pieces
        e.train_stream(u_d[a:b], p_d[a:b], sampler=eng.NEG_UNIFORM, seed=7, offset=(r << 40) + a, max_inflight=1)

    # the real thing: this rank, RCCL inside the library
    e = engine()
    uid = [eng.Engine.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    e.comm_init(uid[0], rank, world)
    e.comm_hot_tier(hot, counts)
    for k in range(rounds):
        for p in range(pieces):
            launch(e, rank, k, p)
            e.hot_sync()
        e.item_sync()
    e.item_sync_finish()
    torch.cuda.synchronize()
    # the same protocol with all ranks in this process
    lw = LocalWorld(world)
    es = [engine() for _ in range(world)]
    ss = [ItemSync([es[r].Q], comm=lw.member(r), engine=es[r], hot_rows=H, item_counts=counts) for r in range(world)]
    for k in range(rounds):
        for p in range(pieces):
            for r in range(world):
                launch(es[r], r, k, p)
                ss[r].hot_step()
        for r in range(world):
            ss[r].step()
    for r in range(world):
        ss[r].hot_finish()
        ss[r].finish()
    torch.cuda.synchronize()
    err = (e.Q - es[rank].Q).abs().max().item()
    errp = (e.P - es[rank].P).abs().max().item()
    moved = (e.Q - torch.from_numpy(Q0).to(dev)).abs().max().item()
    flag = torch.tensor([err, errp], device=dev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(f"two-tier through the C ABI over RCCL, world {world}: max |Q - in-process protocol| = {flag[0].item():.3e}, "
              f"|P| {flag[1].item():.3e} (tolerance 1e-5; the table moved by up to {moved:.3f})")
        assert flag[0].item() < 1e-5 and flag[1].item() < 1e-5
        print("OK two-tier")
    e.comm_destroy()
    for s_ in ss:
        s_.close()


if __name__ == "__main__":
    main()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    if os.environ.get("BPR_DIST_BACKEND", "nccl") == "nccl":  # RCCL inside the library needs a device per rank
        two_tier(world, rank, torch.device("cuda", local))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
