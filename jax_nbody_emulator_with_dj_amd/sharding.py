"""Multi-GPU sharding of process_box: one process per GPU, bricks + halo exchange.

Nothing like this exists in the reference (its loop is serial on one device,
subbox.py:195-215); the decomposition follows SURVEY.md section 8e: the ranks form a
periodic brick grid, every rank keeps its un-haloed brick of the box resident in
HBM, fetches the 48-voxel halo from its neighbours with point-to-point messages
(torch.distributed P2P = RCCL send/recv over xGMI on GPUs, gloo on CPUs), and then runs
the sub-boxes of its own brick.  There is no all-reduce or gather anywhere: weights
are replicated and outputs stay in the owning rank's brick.

The exchange is axis by axis (3 rounds, 2 messages per split axis) so that edges and
corners arrive through the face messages: the slab sent along axis a spans the halos
already filled on axes < a.  Bytes per rank for 512^3 on a 2x2x2 grid: 322 MB.

z-slab bricks (rank grid (N,1,1), the default whenever a slab is at least 48 planes deep).
A brick that is not split in y and x is periodic there by itself, so only its z halo costs
anything -- and almost all of that cost sits BELOW the full-resolution level: the 48-plane halo of
the raw input exists to give levels 1-3 their context.  Instead of recomputing it, the ranks
exchange what each level needs where it is smallest (include/nbe.h, "Brick mode"): 4 planes of raw
input per side (13 MB at 512^2), 6 planes of the down_l0 output (201 MB; the interior of conv_l1
runs while they travel, on the communication stream), 10 planes of the down_l1 output (84 MB), and
4 planes of the level-0 skip connection (541 MB, needed by the decoder only: they travel while
levels 1-3 run).  What is still recomputed is 2 planes per layer and side inside the level-0 blocks,
4 + 4 planes of conv_l1 and the 10-plane halo of levels 2-3 (1.5 % of the work).  The fields equal the single-GPU run
(bit for bit on the direct kernels; to float32 rounding where the Winograd-z kernel pairs planes
differently).  The exchange buffers are allocated once per ShardedBox.  A brick whose workspace does
not fit the memory that is free (all ranks agree on that with a 4-byte all-reduce) falls back to the
padded bricks below.
"""

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

PAD = 48


BRICK_MIN_DEPTH = 48          # a brick hands 10 planes of its down_l1 output (a quarter of its depth) to either neighbour
RAW_HALO = 4                  # planes of raw input a brick needs from either z neighbour


def _zbrick_factor(e0):
    """Work per output voxel of a z-slab brick of depth e0 with the three exchanges, relative to no halo at all:
    level 0 (90 % of the FLOPs) computes ~4 extra planes, level 1 (8.5 %) ~7 extra half-resolution planes, levels 2-3
    (1.5 %) 20 quarter-resolution ones."""
    return 1.0 + 0.90 * 4.0 / e0 + 0.085 * 7.0 / (e0 / 2.0) + 0.015 * 20.0 / (e0 / 4.0)


def _halo_factor(e):
    """Work per output voxel of a padded axis of extent e relative to an axis without halo recompute (fit to the
    measured 17.1 / 11.2 / 9.33 MFLOP per voxel of 128^3 / 256^3 / 512^3 tiles against the network's 7.93)."""
    return 1.0 + 30.0 / e + 900.0 / (e * e)


def rank_grid(world_size, ndiv, size=None, pad=PAD, zbricks=True):
    """Split `world_size` ranks over the sub-box grid `ndiv`: the factorisation with the least halo recompute.
    A brick that is not split in y and x runs the engine's periodic mode there and pays only for its z halo;
    otherwise every axis pays its padded factor.  z-slab bricks of at least 44 planes exchange their level-1 context
    instead of recomputing it (zbricks, see the module docstring).  512^3 / ndiv 4: 2 -> (2,1,1), 4 -> (4,1,1),
    8 -> (8,1,1) when the sub-box grid allows it, else (2,2,2); the sub-box grid only has to be divisible by the rank
    grid in y and x (along z a rank takes its slab as one tile).  Ties go to the leading axes."""
    n = int(world_size)
    if size is None:
        size = tuple(128 * d for d in ndiv)
    best, best_cost = None, None
    for g0 in range(1, n + 1):
        zslab_ok = zbricks and size[0] % g0 == 0 and (size[0] // g0) % 8 == 0 and size[0] // g0 >= BRICK_MIN_DEPTH
        if n % g0 or (ndiv[0] % g0 and not (zslab_ok and g0 == n)):
            continue
        for g1 in range(1, n // g0 + 1):
            if (n // g0) % g1 or ndiv[1] % g1:
                continue
            g2 = n // (g0 * g1)
            if ndiv[2] % g2:
                continue
            e = (size[0] / g0, size[1] / g1, size[2] / g2)
            if any(g > 1 and ee < pad for g, ee in zip((g0, g1, g2), e)):
                continue                                  # a brick must hold its neighbour's halo
            if g1 == 1 and g2 == 1:
                if g0 > 1 and e[0] >= BRICK_MIN_DEPTH and e[0] % 8 == 0 and zbricks:
                    cost = _zbrick_factor(e[0])               # level-1 context exchanged, not recomputed
                else:
                    cost = _halo_factor(e[0]) if g0 > 1 else 1.0
            else:
                cost = _halo_factor(e[0]) * _halo_factor(e[1]) * _halo_factor(e[2])
            key = (round(cost, 6), -g0, -g1)
            if best is None or key < best_cost:
                best, best_cost = (g0, g1, g2), key
    if best is None:
        raise ValueError("cannot distribute %d ranks over a %s sub-box grid" % (world_size, tuple(ndiv)))
    return best


def rank_coords(rank, grid):
    return (rank // (grid[1] * grid[2]), (rank // grid[2]) % grid[1], rank % grid[2])


def coords_rank(c, grid):
    return ((c[0] % grid[0]) * grid[1] + (c[1] % grid[1])) * grid[2] + (c[2] % grid[2])


def brick_extent(coords, grid, size):
    """(origin, shape) of the brick of rank `coords` in a box of spatial `size`."""
    shape = tuple(s // g for s, g in zip(size, grid))
    origin = tuple(c * b for c, b in zip(coords, shape))
    return origin, shape


def exchange_halo(brick, grid, coords, pad=PAD, group=None, pad_unsplit=True):
    """brick: (C, b0, b1, b2) tensor (CPU with gloo, CUDA with nccl).  Returns the haloed
    brick whose halo holds the periodic neighbours' voxels: (C, b0+2p, b1+2p, b2+2p), or with
    pad_unsplit=False haloed only along the axes the rank grid splits -- along the others the brick
    is the periodic box itself and the engine wraps around (and can run its periodic mode)."""
    C, b0, b1, b2 = brick.shape
    b = (b0, b1, b2)
    pa = tuple(pad if (grid[a] > 1 or pad_unsplit) else 0 for a in range(3))
    for a in range(3):
        if pa[a] > 0 and b[a] < pad:                # axes without a halo (unsplit, pad_unsplit=False) may be any size
            raise ValueError("brick extent %d along axis %d is smaller than the halo %d" % (b[a], a, pad))
    H = torch.empty((C, b0 + 2 * pa[0], b1 + 2 * pa[1], b2 + 2 * pa[2]), dtype=brick.dtype, device=brick.device)
    H[:, pa[0]:pa[0] + b0, pa[1]:pa[1] + b1, pa[2]:pa[2] + b2] = brick
    for a in range(3):
        if pa[a] == 0:
            continue
        # extents on the other axes: halos included where already filled (axes < a), centre otherwise
        def sl(lo, hi):
            idx = [slice(None)]
            for k in range(3):
                if k == a:
                    idx.append(slice(lo, hi))
                elif k < a:
                    idx.append(slice(None))
                else:
                    idx.append(slice(pa[k], pa[k] + b[k]))
            return tuple(idx)
        first = sl(pad, 2 * pad)                    # my first `pad` planes  -> minus neighbour's high halo
        last = sl(b[a], b[a] + pad)                 # my last  `pad` planes  -> plus  neighbour's low halo
        lo_halo, hi_halo = sl(0, pad), sl(pad + b[a], 2 * pad + b[a])
        if grid[a] == 1:
            H[hi_halo] = H[first]
            H[lo_halo] = H[last]
            continue
        cm, cp = list(coords), list(coords)
        cm[a] -= 1
        cp[a] += 1
        minus, plus = coords_rank(cm, grid), coords_rank(cp, grid)
        s_first, s_last = H[first].contiguous(), H[last].contiguous()
        # gloo moves host memory: stage CUDA slabs through the CPU (test rigs with several ranks on one GPU);
        # with nccl (= RCCL) the slabs go device to device over xGMI
        stage = brick.is_cuda and dist.get_backend(group) == "gloo"
        if stage:
            s_first, s_last = s_first.cpu(), s_last.cpu()
        r_hi, r_lo = torch.empty_like(s_first), torch.empty_like(s_last)
        # order discipline (P2P between one pair matches in order; minus == plus when grid[a] == 2):
        # sends  (1) first -> minus, (2) last -> plus ; receives (1) high <- plus, (2) low <- minus
        ops = [dist.P2POp(dist.isend, s_first, minus, group), dist.P2POp(dist.isend, s_last, plus, group),
               dist.P2POp(dist.irecv, r_hi, plus, group), dist.P2POp(dist.irecv, r_lo, minus, group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        H[hi_halo] = r_hi
        H[lo_halo] = r_lo
    return H


def local_ndiv(ndiv, grid):
    return tuple(n // g for n, g in zip(ndiv, grid))


def split_interior(nd_local, bshape, pad=PAD):
    """Sub-box indices of a brick whose haloed crops [a-pad, a+c+pad) stay inside the brick
    (no neighbour data needed) and the rest, both in row-major order."""
    crop = tuple(b // n for b, n in zip(bshape, nd_local))
    interior, boundary = [], []
    for idx in range(int(np.prod(nd_local))):
        i = (idx // (nd_local[1] * nd_local[2]), (idx // nd_local[2]) % nd_local[1], idx % nd_local[2])
        inner = all(i[a] * crop[a] - pad >= 0 and (i[a] + 1) * crop[a] + pad <= bshape[a] for a in range(3))
        (interior if inner else boundary).append(idx)
    return interior, boundary


def exchange_z_faces(send_lo, send_hi, recv_lo, recv_hi, coords, grid, group=None):
    """Boundary planes of the z-slab bricks: my low planes become the high halo of my z-minus neighbour, my high planes
    the low halo of my z-plus neighbour (periodic).  Tensors of equal size, CUDA with nccl (= RCCL: device to device
    over xGMI) or CPU / CUDA with gloo (CUDA tensors are staged through the host: test rigs with several ranks per GPU)."""
    if grid[0] == 1:                                           # one brick: its own periodic images
        recv_hi.copy_(send_lo)
        recv_lo.copy_(send_hi)
        return
    cm, cp = list(coords), list(coords)
    cm[0] -= 1
    cp[0] += 1
    minus, plus = coords_rank(cm, grid), coords_rank(cp, grid)
    stage = send_lo.is_cuda and dist.get_backend(group) == "gloo"
    s_lo, s_hi = (send_lo.cpu(), send_hi.cpu()) if stage else (send_lo, send_hi)
    r_hi, r_lo = (torch.empty_like(s_lo), torch.empty_like(s_hi)) if stage else (recv_hi, recv_lo)
    # order discipline (P2P between one pair matches in order; minus == plus when grid[0] == 2):
    # sends (1) low -> minus, (2) high -> plus ; receives (1) high halo <- plus, (2) low halo <- minus
    ops = [dist.P2POp(dist.isend, s_lo, minus, group), dist.P2POp(dist.isend, s_hi, plus, group),
           dist.P2POp(dist.irecv, r_hi, plus, group), dist.P2POp(dist.irecv, r_lo, minus, group)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if stage:
        recv_hi.copy_(r_hi)
        recv_lo.copy_(r_lo)


class ShardedBox:
    """One rank's share of a periodic box: brick resident on this rank's GPU."""

    def __init__(self, engine, size, ndiv, rank, world_size, group=None, comm_stream=None, zbricks=None):
        import os
        self.eng = engine
        self.size, self.ndiv = tuple(size), tuple(ndiv)
        self.rank, self.world = rank, world_size
        if zbricks is None:
            zbricks = os.environ.get("NBE_ZBRICKS", "1") != "0"
        # the brick mode computes every voxel of the box: only where the reference's sub-boxes do (no remainder) and where
        # cutting the box differently gives the same field (crop % 8 == 0)
        exact = all(s % n == 0 and (s // n) % 8 == 0 for s, n in zip(self.size, self.ndiv))
        self.grid = rank_grid(world_size, ndiv, self.size, zbricks=zbricks and exact)
        self.coords = rank_coords(rank, self.grid)
        self.origin, self.bshape = brick_extent(self.coords, self.grid, size)
        self.zbricks = (zbricks and exact and self.grid[0] > 1 and self.grid[1] == 1 and self.grid[2] == 1 and self.bshape[0] % 8 == 0
                        and self.bshape[0] >= BRICK_MIN_DEPTH and self.bshape[1] >= 48 and self.bshape[2] >= 48)
        self.nd_local = tuple(max(1, n // g) for n, g in zip(ndiv, self.grid))
        self.group = group
        self.comm_stream = comm_stream
        self._halo = None                       # (send_lo, send_hi, recv_lo, recv_hi), allocated once
        self.fallback = None                    # optional strict-float32 Engine with the same parameters (see process)
        self._agreed = False                    # the ranks have agreed on brick mode vs padded bricks (first call)
        self.trace = False                      # bracket the compute stream's waits for exchanges with timing events (wait_ms)
        self._waits = []

    def _exchange_async(self, cur, send_lo, send_hi, recv_lo, recv_hi):
        """Face exchange on the communication stream, behind what `cur` has enqueued so far; returns the event the consumer
        waits for (None: the exchange ran on `cur` itself)."""
        comm = self.comm_stream
        if cur is None or comm is None or comm == cur:          # (CPU tensors: the gloo tests of the protocol)
            exchange_z_faces(send_lo, send_hi, recv_lo, recv_hi, self.coords, self.grid, self.group)
            return None
        comm.wait_stream(cur)
        with torch.cuda.stream(comm):
            exchange_z_faces(send_lo, send_hi, recv_lo, recv_hi, self.coords, self.grid, self.group)
            ev = torch.cuda.Event()
            ev.record(comm)
        for t in (send_lo, send_hi, recv_lo, recv_hi):
            t.record_stream(comm)
        return ev

    def _process_zbrick(self, brick, Dz, vel_fac, disp, vel):
        """z-slab brick: three small exchanges instead of a recomputed 48-plane halo (module docstring).  The compute stream
        never waits for a transfer it could not have overlapped: the 6-plane faces of the down_l0 output travel under the
        interior of conv_l1."""
        cur = torch.cuda.current_stream(brick.device) if brick.is_cuda else None
        H = exchange_halo(brick, self.grid, self.coords, RAW_HALO, self.group, pad_unsplit=False)     # raw input, 4 planes in z
        if self._halo is None or self._halo[0].device != brick.device:
            n = [self.eng.brick_halo_bytes(self.bshape, w) for w in (1, 2, 3)]
            self._halo = tuple(torch.empty(k, dtype=torch.uint8, device=brick.device) for k in (n[0],) * 4 + (n[1],) * 4 + (n[2],) * 4)
        s_lo, s_hi, r_lo, r_hi, s2_lo, s2_hi, r2_lo, r2_hi, k_lo, k_hi, q_lo, q_hi = self._halo
        self.eng.brick_encode(H, self.bshape, Dz, vel_fac, s_lo, s_hi, k_lo, k_hi)
        ev = self._exchange_async(cur, s_lo, s_hi, r_lo, r_hi)
        self.eng.brick_interior()                                   # runs while the faces travel
        self._wait(cur, ev, "down_l0 faces")
        self.eng.brick_exchange(r_lo, r_hi, s2_lo, s2_hi)
        ev = self._exchange_async(cur, s2_lo, s2_hi, r2_lo, r2_hi)
        # the skip connection's planes go last on the communication stream (behind the faces the compute stream is waiting
        # for) and are needed last: the engine waits for them after levels 1-3, inside brick_finish
        ev_skip = self._exchange_async(cur, k_lo, k_hi, q_lo, q_hi)
        self._wait(cur, ev, "down_l1 faces")
        self.eng.brick_finish(r2_lo, r2_hi, q_lo, q_hi, Dz, vel_fac, disp, vel, skip_ready=ev_skip)

    def _wait(self, cur, ev, what):
        """The compute stream waits for an exchange.  With self.trace the wait is bracketed by timing events: how long the
        stream actually stood still for it is the gap between them (read by wait_ms() after the step)."""
        if ev is None:
            return
        if self.trace and cur is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur)
            cur.wait_event(ev)
            b.record(cur)
            self._waits.append((what, a, b))
        else:
            cur.wait_event(ev)

    def wait_ms(self):
        """{exchange: ms the compute stream stood still for it}, summed over the steps since the last call (trace=True)."""
        out = {}
        for what, a, b in self._waits:
            b.synchronize()
            out[what] = out.get(what, 0.0) + a.elapsed_time(b)
        self._waits = []
        return out

    def _agree_on_bricks(self, device):
        """A brick whose workspace does not fit the memory that is free now runs as padded bricks instead; every rank must
        take the same path (the exchanges differ), so they agree once, with a 4-byte all-reduce."""
        if self._agreed or not self.zbricks:
            self._agreed = True
            return
        ok = 1 if self.eng.brick_plan(self.bshape) > 0 else 0
        if self.world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if dist.get_backend(self.group) == "gloo" else device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            ok = int(flag.item())
        if not ok:
            import warnings
            warnings.warn("the z-slab brick %s does not fit the device memory that is free on every rank: falling back to padded "
                          "bricks (more halo recompute)" % (self.bshape,), RuntimeWarning)
            self.zbricks = False
        self._agreed = True

    def backend(self):
        """'nccl' (= RCCL, device to device over xGMI) or 'gloo' (host-staged: CPU tests and one-card rigs); None for one rank."""
        return dist.get_backend(self.group) if self.world > 1 else None

    def process(self, brick, Dz, vel_fac, disp, vel, check_finite=True):
        """brick, disp, vel: CUDA tensors (C, *bshape).  Interior sub-boxes run while the halo
        messages are in flight on the communication stream; boundary sub-boxes wait for them.

        Range (include/nbe.h): all ranks compute with ONE range shift (a 4-byte MAX all-reduce of max|x|), and they agree
        on the outcome as well: when an activation leaves the f16 range on ANY rank, every rank learns it through a second
        4-byte all-reduce and all of them either recompute the step on `fallback` (strict float32 engines, same
        parameters; set ShardedBox.fallback) or raise NBERangeError together -- no rank walks into the next collective
        alone.  The preset range is cleared again before this returns, whatever happens."""
        from .engine import NBERangeError
        try:
            self._process(self.eng, brick, Dz, vel_fac, disp, vel)
            bad, msg = 0, ""
            if check_finite:
                try:
                    self.eng.check_finite()
                except NBERangeError as e:
                    bad, msg = 1, str(e)
        finally:
            self.eng.set_input_range(None)
        if not check_finite:
            return disp, vel
        if self.world > 1:
            flag = torch.tensor([bad], dtype=torch.int32, device="cpu" if dist.get_backend(self.group) == "gloo" else brick.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            anybad = int(flag.item())
        else:
            anybad = bad
        if anybad:
            if self.fallback is None:
                raise NBERangeError(msg or "non-finite values on another rank of the sharded box: an activation left the range "
                                           "of the f16-based arithmetic; rerun the step with strict float32 engines")
            import warnings
            warnings.warn("%s -- recomputing this box with the strict float32 engines on all ranks"
                          % (msg or "an activation left the f16 range on another rank"), RuntimeWarning)
            try:
                self._process(self.fallback, brick, Dz, vel_fac, disp, vel)
                self.fallback.check_finite()
            finally:
                self.fallback.set_input_range(None)
        return disp, vel

    def _process(self, eng, brick, Dz, vel_fac, disp, vel):
        self.eng, keep = eng, self.eng
        try:
            return self._process_on(brick, Dz, vel_fac, disp, vel)
        finally:
            self.eng = keep

    def _process_on(self, brick, Dz, vel_fac, disp, vel):
        cur = torch.cuda.current_stream(brick.device) if brick.is_cuda else None
        self._agree_on_bricks(brick.device)
        # one range shift for the whole box (include/nbe.h, "Range"): max |x| over all bricks, a 4-byte all-reduce --
        # every rank then computes its brick with the arithmetic a single-GPU run of the box would use
        amax = torch.linalg.vector_norm(brick.reshape(-1), ord=float('inf')).float().reshape(1)
        if self.world > 1:
            if dist.get_backend(self.group) == "gloo":
                amax = amax.cpu()
            dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=self.group)
        self.eng.set_input_range(float(amax.item()))
        if self.zbricks:
            self._process_zbrick(brick, Dz, vel_fac, disp, vel)
            return disp, vel
        # the engine merges sub-boxes into larger tiles when that is exact (nbe_plan_tiles); split on that grid
        nd = self.eng.plan_tiles(self.bshape, self.nd_local, periodic_box=False)
        interior, boundary = split_interior(nd, self.bshape)
        if self.comm_stream is not None and interior:
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                H = exchange_halo(brick, self.grid, self.coords, PAD, self.group, pad_unsplit=False)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            # interior crops read the un-haloed brick itself (origin 0): independent of H
            self.eng.process_region(brick, (0, 0, 0), self.bshape, nd, Dz, vel_fac, disp, vel, order=interior)
            cur.wait_event(done)
            H.record_stream(cur)
        else:
            H = exchange_halo(brick, self.grid, self.coords, PAD, self.group, pad_unsplit=False)
            boundary = interior + boundary
        # haloed only along the split axes: along the others the region is the periodic box itself
        origin = tuple(PAD if g > 1 else 0 for g in self.grid)
        self.eng.process_region(H, origin, self.bshape, nd, Dz, vel_fac, disp, vel, order=sorted(boundary))
        return disp, vel
