"""GPU: the N > 1 path end to end on ONE card.  RCCL refuses several ranks on one device, so the ranks
are separate processes that share cuda:0 and exchange their halos over gloo (CPU-staged); everything
else -- brick bookkeeping, halo content, nbe_process_region on the haloed brick, internal tile
merging per brick -- is the code the multi-GPU bench runs.  The assembled result must equal the
single-process process_box of the whole box."""

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Z, OM = 0.5, 0.3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, size, ndiv, seed_p, seed_x, q, zb=True):
    import torch
    os.environ["NBE_ZBRICKS"] = "1" if zb else "0"
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NBE_MEM_FRACTION"] = "%.3f" % (0.8 / world)     # the ranks share one card: plan tiles with a share of it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import jax_nbody_emulator_with_dj_amd as J
        from jax_nbody_emulator_with_dj_amd import sharding
        from jax_nbody_emulator_with_dj_amd.engine import Engine
        from oracle import params as P
        torch.cuda.set_device(0)
        p = P.synthetic_params(seed=seed_p, mid_chan=8)
        eng = Engine(device=0, mid_chan=8, compute_vel=True)
        eng.load_params(p, premodulated=False)
        Dz, vf = float(J.growth_factor(Z, OM)), float(J.vel_norm(Z, OM))
        eng.set_cosmology(OM, Dz)
        full = np.random.default_rng(seed_x).standard_normal((3,) + size).astype(np.float32)
        sb = sharding.ShardedBox(eng, size, ndiv, rank, world)
        o, b = sb.origin, sb.bshape
        want_z = zb and sb.grid[0] > 1 and sb.grid[1] == 1 and sb.grid[2] == 1 and b[0] >= 48
        assert sb.zbricks == want_z, (sb.grid, b, sb.zbricks)
        brick = torch.from_numpy(np.ascontiguousarray(
            full[:, o[0]:o[0] + b[0], o[1]:o[1] + b[1], o[2]:o[2] + b[2]])).cuda()
        disp, vel = torch.zeros_like(brick), torch.zeros_like(brick)
        sb.process(brick, Dz, vf, disp, vel)
        torch.cuda.synchronize()
        q.put((rank, o, disp.cpu().numpy(), vel.cpu().numpy()))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,ndiv", [(2, (128, 64, 64), (2, 1, 1)),         # (2,1,1): 64-plane z-slab bricks, three small exchanges
                                             (4, (128, 128, 64), (4, 2, 1)),        # rank grid (2,2,1): y split, padded
                                             (4, (256, 64, 64), (4, 1, 1)),         # (4,1,1): z-slab bricks
                                             (4, (192, 48, 56), (2, 1, 1)),         # 48-plane bricks; the sub-box grid does not divide by 4
                                             (-4, (256, 64, 64), (4, 1, 1))])       # NBE_ZBRICKS=0: the padded z-slab bricks of round 1
def test_sharded_equals_single_process(world, size, ndiv):
    zb = world > 0
    world = abs(world)
    import torch.multiprocessing as mp
    import jax_nbody_emulator_with_dj_amd as J
    from oracle import params as P
    J.models.release_engines()                                  # give the card back before the rank processes start
    seed_p, seed_x = 61, 62
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, size, ndiv, seed_p, seed_x, q, zb)) for r in range(world)]
    for pr in procs:
        pr.start()
    parts = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    # single-process reference through the plain API
    p = P.synthetic_params(seed=seed_p, mid_chan=8)
    full = np.random.default_rng(seed_x).standard_normal((3,) + size).astype(np.float32)
    proc = J.SubboxProcessor(J.StyleNBodyEmulatorVelCore(mid_chan=8), p, J.SubboxConfig(size=size, ndiv=ndiv))
    d_ref, v_ref = proc.process_box(full, Z, OM, show_progress=False)
    d_all, v_all = np.zeros_like(d_ref), np.zeros_like(v_ref)
    cover = np.zeros(size, np.int32)
    for rank, o, d, v in parts:
        b = d.shape[1:]
        sl = (slice(None), slice(o[0], o[0] + b[0]), slice(o[1], o[1] + b[1]), slice(o[2], o[2] + b[2]))
        d_all[sl], v_all[sl] = d, v
        cover[sl[1:]] += 1
    assert np.all(cover == 1)
    # identical arithmetic per voxel (same kernels, same K order): equal to rounding of the tile boundaries
    np.testing.assert_allclose(d_all, d_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v_all, v_ref, rtol=1e-5, atol=1e-4)
    from jax_nbody_emulator_with_dj_amd import sharding
    grid = sharding.rank_grid(world, ndiv, size)
    if zb and grid[1] == 1 and grid[2] == 1 and size[0] // grid[0] >= 48:
        # z-slab bricks run the single-GPU schedule of the whole box, cut along z, with the same pairing of planes: bit for bit
        assert np.array_equal(d_all, d_ref) and np.array_equal(v_all, v_ref)


def test_brick_calls_guard_their_state(engine_factory):
    """The C ABI's brick calls on one context (include/nbe.h, "Brick mode"): a brick survives only until another call uses
    the context's workspace; the calls must come in order; and faces computed with another range shift are refused (the
    sender's shift travels in the last word of either face: a sharded box whose ranks did not agree on one max |x|)."""
    import torch
    from jax_nbody_emulator_with_dj_amd.engine import NBEError
    from oracle import params as P
    Dz, vf = 0.7731811501855036, 50.537651303131064
    e = engine_factory(mid_chan=8, compute_vel=True, precision="f16x3")
    e.load_params(P.synthetic_params(seed=61, mid_chan=8), premodulated=False)
    e.set_cosmology(OM, Dz)
    b = (64, 64, 64)
    H = torch.randn((3, b[0] + 8, b[1], b[2]), device="cuda")
    n = [e.brick_halo_bytes(b, w) for w in (1, 2, 3)]
    f1 = [torch.zeros(n[0], dtype=torch.uint8, device="cuda") for _ in range(4)]
    f2 = [torch.zeros(n[1], dtype=torch.uint8, device="cuda") for _ in range(4)]
    f3 = [torch.zeros(n[2], dtype=torch.uint8, device="cuda") for _ in range(4)]
    disp, vel = torch.zeros((3,) + b, device="cuda"), torch.zeros((3,) + b, device="cuda")

    def whole(faces_from=None):
        e.brick_encode(H, b, Dz, vf, f1[0], f1[1], f3[0], f3[1])
        f1[2].copy_(f1[1] if faces_from is None else faces_from[1]); f1[3].copy_(f1[0] if faces_from is None else faces_from[0])
        f3[2].copy_(f3[1]); f3[3].copy_(f3[0])
        e.brick_interior()
        e.brick_exchange(f1[2], f1[3], f2[0], f2[1])
        f2[2].copy_(f2[1]); f2[3].copy_(f2[0])
        e.brick_finish(f2[2], f2[3], f3[2], f3[3], Dz, vf, disp, vel)          # (no event: the planes are there)

    try:
        e.set_input_range(4.0)
        whole(); e.check_finite()                                    # a brick that is its own neighbour: fine
        assert bool(torch.isfinite(disp).all()) and float(disp.abs().max()) > 0
        # out of order, and after another call has used the workspace
        with pytest.raises(NBEError, match="brick call"):
            e.brick_interior()
        e.brick_encode(H, b, Dz, vf, f1[0], f1[1], f3[0], f3[1])
        with pytest.raises(NBEError, match="out of order"):
            e.brick_exchange(f1[2], f1[3], f2[0], f2[1])
        e.brick_encode(H, b, Dz, vf, f1[0], f1[1], f3[0], f3[1])
        e.forward(torch.randn((3, 104, 104, 104), device="cuda"), Dz, vf)
        with pytest.raises(NBEError, match="brick call"):
            e.brick_interior()
        # faces from a rank that used another range shift
        other = [f1[0].clone(), f1[1].clone()]
        e.set_input_range(4000.0)
        whole(faces_from=other)
        with pytest.raises(NBEError, match="another range shift"):
            e.check_finite()
    finally:
        e.set_input_range(None)


def test_brick_protocol_over_rccl_with_one_rank_as_its_own_neighbour():
    """RCCL refuses several ranks per device, so what can run on the nccl backend here is ONE rank: the process group, the
    4-byte all-reduces, the grouped P2P of the exchanges to itself, and the whole brick step of ShardedBox -- four face exchanges
    on the communication stream, events into the engine's stream -- with this rank as both of its z neighbours; a brick that is
    its own neighbour is the periodic box, so the fields must equal process_box bit for bit (tools/gpu/rccl_self_check.py, in a
    process of its own: the nccl process group must not meet the gloo groups of the other tests)."""
    import subprocess
    import sys
    import jax_nbody_emulator_with_dj_amd as J
    J.models.release_engines()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu", "rccl_self_check.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and "rccl self check: ok" in r.stdout and "bit-identical: True" in r.stdout
