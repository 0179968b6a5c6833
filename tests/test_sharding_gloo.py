"""CPU, multi-process: the N > 1 path's halo exchange and brick bookkeeping with the gloo backend
(world_size 2 and 4 on the CPU); the same code runs over RCCL on GPUs."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jax_nbody_emulator_with_dj_amd import sharding as S


def test_rank_grid_and_bricks():
    assert S.rank_grid(1, (4, 4, 4)) == (1, 1, 1)
    assert S.rank_grid(2, (4, 4, 4)) == (2, 1, 1)
    assert S.rank_grid(4, (4, 4, 4)) == (4, 1, 1)             # z-only split: the bricks stay periodic in y and x
    assert S.rank_grid(4, (2, 4, 4), zbricks=False) == (1, 2, 2)   # padded bricks: cubes beat slabs once every axis is padded anyway
    assert S.rank_grid(4, (2, 4, 4)) == (4, 1, 1)             # z-slab bricks need not follow the sub-box grid along z
    # z-slab bricks that exchange their level-1 context (sharding.py): slabs of >= 44 planes beat cubes
    assert S.rank_grid(8, (4, 4, 4)) == (8, 1, 1) and S.rank_grid(8, (4, 4, 4), zbricks=False) == (2, 2, 2)
    assert S.rank_grid(8, (8, 8, 8)) == (8, 1, 1) and S.rank_grid(8, (8, 8, 8), zbricks=False) == (2, 2, 2)
    assert S.rank_grid(8, (4, 4, 4), (256,) * 3) == (2, 2, 2)          # 32-plane slabs are too thin for brick mode (a brick hands a quarter of its depth, 10 planes, to either neighbour)
    assert S._zbrick_factor(64) < 1.3 < S._halo_factor(256) ** 3
    assert S.rank_grid(16, (4, 4, 4)) == (4, 2, 2)
    assert S.rank_grid(2, (1, 4, 4), zbricks=False) == (1, 2, 1) and S.rank_grid(2, (1, 4, 4)) == (2, 1, 1)
    with pytest.raises(ValueError):
        S.rank_grid(8, (1, 1, 4))
    grid = (2, 2, 2)
    seen = set()
    cover = np.zeros((512,) * 3, np.int8)
    for r in range(8):
        c = S.rank_coords(r, grid)
        assert S.coords_rank(c, grid) == r
        o, b = S.brick_extent(c, grid, (512,) * 3)
        assert b == (256, 256, 256)
        cover[o[0]:o[0] + b[0], o[1]:o[1] + b[1], o[2]:o[2] + b[2]] += 1
        seen.add(c)
    assert len(seen) == 8 and np.all(cover == 1)
    assert S.coords_rank((-1, 2, 0), grid) == S.coords_rank((1, 0, 0), grid)     # periodic
    assert S.local_ndiv((4, 4, 4), grid) == (2, 2, 2)


def test_split_interior():
    # C4: 2x2x2 sub-boxes of 128 per 256-brick -> every crop needs the halo
    i, b = S.split_interior((2, 2, 2), (256, 256, 256))
    assert i == [] and len(b) == 8
    # C5: 4x4x4 sub-boxes of 128 per 512-brick -> the inner 2x2x2 are independent of the neighbours
    i, b = S.split_interior((4, 4, 4), (512, 512, 512))
    assert len(i) == 8 and len(b) == 56 and sorted(i + b) == list(range(64))
    assert i[0] == (1 * 4 + 1) * 4 + 1
    # small crops (crop < pad): an "inner" index is not enough, the haloed crop must fit
    i, b = S.split_interior((4, 4, 4), (128, 128, 128))
    assert i == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ndiv, size, pad, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = S.rank_grid(world, ndiv, size, pad)
        coords = S.rank_coords(rank, grid)
        origin, bshape = S.brick_extent(coords, grid, size)
        full = torch.from_numpy(np.random.default_rng(123).standard_normal((3,) + size).astype(np.float32))
        brick = full[:, origin[0]:origin[0] + bshape[0], origin[1]:origin[1] + bshape[1],
                     origin[2]:origin[2] + bshape[2]].contiguous()
        H = S.exchange_halo(brick, grid, coords, pad)
        # reference: periodic gather of the haloed brick from the global box (subbox.py:90-95 index rule)
        idx = [np.arange(o - pad, o + b + pad) % s for o, b, s in zip(origin, bshape, size)]
        want = full[:, idx[0][:, None, None], idx[1][None, :, None], idx[2][None, None, :]]
        ok = bool(torch.equal(H, want))
        # haloed only along the axes the rank grid splits (what ShardedBox hands to the engine)
        H2 = S.exchange_halo(brick, grid, coords, pad, pad_unsplit=False)
        pa = [pad if g > 1 else 0 for g in grid]
        idx = [np.arange(o - p, o + b + p) % s for o, b, s, p in zip(origin, bshape, size, pa)]
        want2 = full[:, idx[0][:, None, None], idx[1][None, :, None], idx[2][None, None, :]]
        ok = ok and bool(torch.equal(H2, want2))
        q.put((rank, ok, tuple(H.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ndiv,size,pad", [(2, (2, 2, 2), (16, 12, 10), 4),
                                                 (2, (1, 2, 1), (10, 16, 12), 5),
                                                 (4, (4, 4, 4), (24, 16, 12), 6),
                                                 (4, (4, 1, 1), (32, 10, 8), 8)])
def test_halo_exchange_matches_periodic_gather(world, ndiv, size, pad):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ndiv, size, pad, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res


def _zface_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = (world, 1, 1)
        coords = S.rank_coords(rank, grid)
        n = 1000 + 7 * rank                                        # every rank must size its buffers alike: use the max
        n = 1021
        s_lo = torch.full((n,), 10 * rank + 1, dtype=torch.uint8)
        s_hi = torch.full((n,), 10 * rank + 2, dtype=torch.uint8)
        r_lo, r_hi = torch.zeros(n, dtype=torch.uint8), torch.zeros(n, dtype=torch.uint8)
        S.exchange_z_faces(s_lo, s_hi, r_lo, r_hi, coords, grid)
        minus, plus = (rank - 1) % world, (rank + 1) % world
        # my low halo is the z-minus neighbour's high planes, my high halo the z-plus neighbour's low planes
        ok = bool((r_lo == 10 * minus + 2).all()) and bool((r_hi == 10 * plus + 1).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_z_face_exchange(world):
    """A face exchange of the brick mode: every rank's boundary planes land in the right neighbour's halo, also when
    both neighbours are the same rank (world 2).  World 1 is a local periodic copy."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zface_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
    a, b = torch.arange(5, dtype=torch.uint8), torch.arange(5, 10, dtype=torch.uint8)
    ra, rb = torch.zeros(5, dtype=torch.uint8), torch.zeros(5, dtype=torch.uint8)
    S.exchange_z_faces(a, b, ra, rb, (0, 0, 0), (1, 1, 1))
    assert torch.equal(ra, b) and torch.equal(rb, a)


class _FakeEngine:
    """Stands in for Engine in the CPU test of ShardedBox's brick protocol: records the calls, marks the faces it hands
    out with its rank, checks the faces it receives, and can pretend that an activation left the f16 range."""
    RAW_HALO = 4

    def __init__(self, rank, world, bad=False):
        self.rank, self.world, self.bad, self.calls, self.range = rank, world, bad, [], "unset"

    def set_input_range(self, a):
        self.range = a

    def brick_plan(self, b):
        return 32

    def brick_halo_bytes(self, b, which):
        return {1: 64, 2: 32, 3: 48}[which]

    def brick_encode(self, H, b, Dz, vf, s_lo, s_hi, k_lo, k_hi):
        assert tuple(H.shape) == (3, b[0] + 8, b[1], b[2]) and self.range is not None and self.range != "unset"
        for t, v in ((s_lo, 1), (s_hi, 2), (k_lo, 5), (k_hi, 6)):
            t.fill_(10 * self.rank + v)
        self.calls.append("encode")

    def brick_interior(self):
        self.calls.append("interior")

    def brick_exchange(self, r_lo, r_hi, s2_lo, s2_hi):
        minus, plus = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        assert bool((r_lo == 10 * minus + 2).all()) and bool((r_hi == 10 * plus + 1).all())
        s2_lo.fill_(10 * self.rank + 3); s2_hi.fill_(10 * self.rank + 4)
        self.calls.append("exchange")

    def brick_finish(self, r2_lo, r2_hi, q_lo, q_hi, Dz, vf, disp, vel, skip_ready=None):
        minus, plus = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        assert bool((r2_lo == 10 * minus + 4).all()) and bool((r2_hi == 10 * plus + 3).all())
        assert bool((q_lo == 10 * minus + 6).all()) and bool((q_hi == 10 * plus + 5).all())
        disp.fill_(float(self.rank)); vel.fill_(float(self.rank))
        self.calls.append("finish")

    def check_finite(self):
        from jax_nbody_emulator_with_dj_amd.engine import NBERangeError
        if self.bad:
            raise NBERangeError("non-finite values in the output of a finite input (fake)")


def _protocol_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        from jax_nbody_emulator_with_dj_amd.engine import NBERangeError
        size, ndiv = (48 * world, 48, 48), (world, 1, 1)
        brick = torch.full((3, 48, 48, 48), float(rank + 1))
        disp, vel = torch.zeros_like(brick), torch.zeros_like(brick)
        out = {}
        # 1. a clean step: the four brick calls in order, one range for all ranks, cleared afterwards
        e = _FakeEngine(rank, world)
        sb = S.ShardedBox(e, size, ndiv, rank, world)
        assert sb.zbricks and sb.grid == (world, 1, 1)
        sb.process(brick, 0.77, 50.0, disp, vel)
        out["clean"] = e.calls == ["encode", "interior", "exchange", "finish"] and e.range is None and float(disp[0, 0, 0, 0]) == rank
        # 2. rank 1 overflows: EVERY rank raises, none is left in a collective, and the preset range is cleared
        e = _FakeEngine(rank, world, bad=(rank == 1))
        sb = S.ShardedBox(e, size, ndiv, rank, world)
        try:
            sb.process(brick, 0.77, 50.0, disp, vel)
            out["raise"] = False
        except NBERangeError:
            out["raise"] = e.range is None
        # 3. with strict-float32 engines as fallback every rank recomputes, together
        e = _FakeEngine(rank, world, bad=(rank == 1))
        fb = _FakeEngine(rank, world)
        sb = S.ShardedBox(e, size, ndiv, rank, world)
        sb.fallback = fb
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sb.process(brick, 0.77, 50.0, disp, vel)
        out["fallback"] = fb.calls == ["encode", "interior", "exchange", "finish"] and fb.range is None and e.range is None
        dist.barrier()                                             # nobody is stuck
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_brick_protocol_and_collective_error_handling(world):
    """ShardedBox with z-slab bricks on CPU tensors and a fake engine: the call order and the routing of all four exchanges
    (also world 2, where both neighbours are one rank), the box-wide range preset being cleared after every step, and the
    ranks agreeing on a range error: all raise, or all recompute on their fallback engines -- nobody hangs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_protocol_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert out == {"clean": True, "raise": True, "fallback": True}, (rank, out)
