"""One rank's share of an N-GPU run of the 512^3 box, timed on one card.

  z-slab bricks (rank grid (N,1,1), sharding.py): periodic in y and x; per box three exchanges with either z neighbour --
  4 planes of raw input, 6 planes of the down_l0 output (the interior of conv_l1 runs meanwhile), 10 planes of the down_l1
  output, 4 planes of the level-0 skip connection (under levels 1-3) -- here copied from the brick's own send buffers on a second stream (same bytes, same kernels, same stream
  choreography as sharding.ShardedBox._process_zbrick, no link) -- Engine.brick_encode / brick_interior / brick_exchange /
  brick_finish.
  padded bricks (the round-1 scheme, NBE_ZBRICKS=0): haloed along the axes the rank grid splits, halo recomputed.

  --link GB/s   make every exchange take as long on the communication stream as that link rate would (a spin kernel behind
                the copy): the timeline a real xGMI hop produces -- whether the compute stream ever waits for a transfer shows
                as the difference to the run without it.  --link 0 (default): copies only.
  --box 1024    one rank's z-slab brick of the 1024^3 box on eight cards (BASELINE config 5): (128, 1024, 1024) with the
                exchanges of that size (the whole box does not fit one card as one tile: N = 8 only, efficiency against
                eight times the brick's own compute is not printed)."""
import sys, time
LINK = float(sys.argv[sys.argv.index("--link") + 1]) if "--link" in sys.argv else 0.0
BOX = int(sys.argv[sys.argv.index("--box") + 1]) if "--box" in sys.argv else 512
sys.path.insert(0, ".")
import torch
from jax_nbody_emulator_with_dj_amd.engine import Engine
from jax_nbody_emulator_with_dj_amd import StyleNBodyEmulatorVelCore

e = Engine(device=0)
e.load_params(StyleNBodyEmulatorVelCore().init(1), False)
e.set_cosmology(0.3, 0.77)
N = BOX
t1 = None
for n in ((1, 2, 4, 8) if N <= 512 else (8,)):
    b = (N // n, N, N)
    if n == 1:
        box = torch.randn((3,) + b, device="cuda")
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d, v = e.process_box(box, b, (4, 4, 4), ((48, 48),) * 3, 0.77, 50.0)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t1 = dt
        print("N=1 whole box, one periodic tile: %.3f s -> %.1f Mvox/s" % (dt, N ** 3 / dt / 1e6), flush=True)
        del box, d, v
        continue
    H = torch.randn((3, b[0] + 8, N, N), device="cuda")
    disp = torch.zeros((3,) + b, device="cuda"); vel = torch.zeros_like(disp)
    n1, n2 = e.brick_halo_bytes(b, 1), e.brick_halo_bytes(b, 2)
    s_lo, s_hi, r_lo, r_hi = (torch.empty(n1, dtype=torch.uint8, device="cuda") for _ in range(4))
    s2_lo, s2_hi, r2_lo, r2_hi = (torch.empty(n2, dtype=torch.uint8, device="cuda") for _ in range(4))
    n3 = e.brick_halo_bytes(b, 3)
    k_lo, k_hi, q_lo, q_hi = (torch.empty(n3, dtype=torch.uint8, device="cuda") for _ in range(4))
    cur, comm = torch.cuda.current_stream(), torch.cuda.Stream()

    def exchange(a_lo, a_hi, b_lo, b_hi):
        comm.wait_stream(cur)
        with torch.cuda.stream(comm):
            b_hi.copy_(a_lo); b_lo.copy_(a_hi)
            if LINK > 0:                                           # both directions travel at once, each at the link rate
                torch.cuda._sleep(int(a_lo.numel() / (LINK * 1e9) * 2.0e9))     # cycles of a ~2 GHz counter
            ev = torch.cuda.Event(); ev.record(comm)
        return ev

    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.brick_encode(H, b, 0.77, 50.0, s_lo, s_hi, k_lo, k_hi)
        ev = exchange(s_lo, s_hi, r_lo, r_hi)
        e.brick_interior()
        cur.wait_event(ev)
        e.brick_exchange(r_lo, r_hi, s2_lo, s2_hi)
        ev2 = exchange(s2_lo, s2_hi, r2_lo, r2_hi)
        ev_skip = exchange(k_lo, k_hi, q_lo, q_hi)               # last on the communication stream, needed last
        cur.wait_event(ev2)
        e.brick_finish(r2_lo, r2_hi, q_lo, q_hi, 0.77, 50.0, disp, vel, skip_ready=ev_skip)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(("link emulated at %g GB/s: " % LINK if LINK > 0 else "") + "N=%d z-slab brick %s, exchanges of %.0f + %.0f + %.0f + %.0f MB per direction: %.3f s -> %.1f Mvox/s for the job, efficiency %.2f"
          % (n, b, e.brick_halo_bytes(b, 0) / 1e6, n1 / 1e6, n2 / 1e6, n3 / 1e6, dt, N ** 3 / dt / 1e6, t1 / (n * dt) if t1 else float("nan")), flush=True)
    del H, disp, vel, s_lo, s_hi, r_lo, r_hi, s2_lo, s2_hi, r2_lo, r2_hi, k_lo, k_hi, q_lo, q_hi
    torch.cuda.empty_cache()
# the padded scheme of round 1 for comparison
for name, grid in [] if (LINK > 0 or N > 512) else (("N=8 (2,2,2)", (2, 2, 2)), ("N=4 (4,1,1)", (4, 1, 1))):
    b = tuple(N // g for g in grid)
    pa = tuple(48 if g > 1 else 0 for g in grid)
    H = torch.randn((3,) + tuple(bb + 2 * p for bb, p in zip(b, pa)), device="cuda")
    disp = torch.zeros((3,) + b, device="cuda"); vel = torch.zeros_like(disp)
    nd_local = tuple(4 // g for g in grid)
    nd = e.plan_tiles(b, nd_local, periodic_box=False)
    order = list(range(nd[0] * nd[1] * nd[2]))
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.process_region(H, pa, b, nd, 0.77, 50.0, disp, vel, order=order)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = grid[0] * grid[1] * grid[2]
    print("%s padded brick %s tiles %s: %.3f s -> %.1f Mvox/s for the job, efficiency %.2f"
          % (name, b, nd, dt, N ** 3 / dt / 1e6, t1 / (n * dt)), flush=True)
    del H, disp, vel
