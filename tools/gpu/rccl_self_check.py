"""RCCL rehearsal on a ONE-GPU box: what bench.py --gpus N / ShardedBox do with torch.distributed, on the "nccl" backend with
world_size 1 -- process-group creation with a device id, the 4-byte MAX all-reduce of the range shift, barrier, and the
grouped P2P pattern of the halo exchanges (batch_isend_irecv with both neighbours being this rank: send to self / receive
from self inside one group, as at world_size 2 where minus == plus), CUDA tensors, side communication stream.

RCCL refuses several ranks per device, so N > 1 cannot run here; this checks that the call sequence itself is accepted by RCCL
and ordered correctly against the compute stream.  Run:  python tools/gpu/rccl_self_check.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jax_nbody_emulator_with_dj_amd import sharding          # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print("backend", dist.get_backend(), "world", dist.get_world_size())
    a = torch.tensor([3.5], device=dev)
    dist.all_reduce(a, op=dist.ReduceOp.MAX)
    assert float(a.item()) == 3.5
    dist.barrier()
    # the z-face exchange of the brick mode on a (2,1,1)-shaped pattern collapsed onto one rank: minus == plus == me
    n = 1 << 20
    s_lo = torch.arange(n, device=dev, dtype=torch.float32)
    s_hi = -torch.arange(n, device=dev, dtype=torch.float32)
    r_lo, r_hi = torch.empty_like(s_lo), torch.empty_like(s_hi)
    ops = [dist.P2POp(dist.isend, s_lo, 0), dist.P2POp(dist.isend, s_hi, 0),
           dist.P2POp(dist.irecv, r_hi, 0), dist.P2POp(dist.irecv, r_lo, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    # order discipline of exchange_z_faces: receive (1) takes send (1), receive (2) takes send (2)
    assert torch.equal(r_hi, s_lo) and torch.equal(r_lo, s_hi), "P2P matching order differs from sharding.exchange_z_faces"
    # the same on a side stream while the compute stream is busy (ShardedBox.process: comm_stream + event)
    comm = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    big = torch.randn(4096, 4096, device=dev)
    for _ in range(4):
        big = big @ big * 1e-3
    comm.wait_stream(cur)
    with torch.cuda.stream(comm):
        H = sharding.exchange_halo(torch.randn(3, 64, 48, 48, device=dev), (1, 1, 1), (0, 0, 0), 48, None, pad_unsplit=True)
        done = torch.cuda.Event()
        done.record(comm)
    cur.wait_event(done)
    H.record_stream(cur)
    assert H.shape == (3, 160, 144, 144) and bool(torch.isfinite(H).all())
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("rccl self check: ok")


if __name__ == "__main__":
    main()
