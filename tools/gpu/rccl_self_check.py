"""RCCL rehearsal on a ONE-GPU box: what bench.py --gpus N / ShardedBox do with torch.distributed, on the "nccl" backend with
world_size 1 -- process-group creation with a device id, the 4-byte MAX all-reduce of the range shift, barrier, and the
grouped P2P pattern of the halo exchanges (batch_isend_irecv with both neighbours being this rank: send to self / receive
from self inside one group, as at world_size 2 where minus == plus), CUDA tensors, side communication stream.

RCCL refuses several ranks per device, so N > 1 cannot run here; this checks that the call sequence itself is accepted by RCCL
and ordered correctly against the compute stream -- and then runs the whole brick step of ShardedBox over RCCL with this rank as
both of its own z neighbours (brick_protocol_with_itself_as_neighbour), against process_box of the same box.  Run:  python tools/gpu/rccl_self_check.py"""
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
    brick_protocol_with_itself_as_neighbour(dev)
    # the float16 model at a width where its Winograd-z form and fused skips apply (Cin a multiple of 32)
    brick_protocol_with_itself_as_neighbour(dev, precision="f16", mid_chan=32)
    dist.destroy_process_group()
    print("rccl self check: ok")


def brick_protocol_with_itself_as_neighbour(dev, precision="f16x3", mid_chan=8):
    """The whole z-slab brick step of ShardedBox over RCCL -- four face exchanges as grouped P2P on the communication stream,
    events into the engine's stream, the skip-connection planes waited for inside nbe_brick_finish -- with ONE rank that is its
    own z-minus and z-plus neighbour (a rank grid of (2,1,1) whose two ranks are this process: what world_size 2 does, where
    minus == plus as well).  A brick that is its own neighbour is the periodic box, so the fields must be those of process_box."""
    from jax_nbody_emulator_with_dj_amd.engine import Engine
    from oracle import params as P
    size = (64, 64, 64)
    eng = Engine(device=0, mid_chan=mid_chan, compute_vel=True, precision=precision)
    eng.load_params(P.synthetic_params(seed=61, mid_chan=mid_chan), premodulated=False)
    Dz, vf = 0.7731811501855036, 50.537651303131064
    eng.set_cosmology(0.3, Dz)
    box = torch.randn((3,) + size, device=dev)
    d_ref, v_ref = eng.process_box(box, size, (1, 1, 1), ((48, 48),) * 3, Dz, vf)
    torch.cuda.synchronize()
    sb = sharding.ShardedBox(eng, size, (1, 1, 1), 0, 1, comm_stream=torch.cuda.Stream(device=dev))
    sb.grid, sb.coords, sb.bshape, sb.zbricks, sb._agreed = (2, 1, 1), (0, 0, 0), size, True, True
    keep = sharding.coords_rank
    sharding.coords_rank = lambda c, g: 0                          # both neighbours are this rank
    try:
        disp, vel = torch.zeros_like(box), torch.zeros_like(box)
        for _ in range(2):
            sb.process(box, Dz, vf, disp, vel)
    finally:
        sharding.coords_rank = keep
    torch.cuda.synchronize()
    ed, ev = float((disp - d_ref).abs().max()), float((vel - v_ref).abs().max())
    print("brick protocol over RCCL, own neighbour, %s mid_chan %d: max|delta| disp %.3g vel %.3g (bit-identical: %s)"
          % (precision, mid_chan, ed, ev, bool(torch.equal(disp, d_ref) and torch.equal(vel, v_ref))))
    assert torch.equal(disp, d_ref) and torch.equal(vel, v_ref)
    eng.close()


if __name__ == "__main__":
    main()
