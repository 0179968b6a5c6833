#!/usr/bin/env python
"""Headline benchmark: output voxels/s of process_box (displacement + velocity) on a 512^3 box,
ndiv=(4,4,4), StyleNBodyEmulatorVelCore, float32, on N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of process_box over the whole box (64 sub-boxes of 224^3 -> 128^3, which the engine
merges into the largest tiles whose workspace fits the card -- ONE tile on a free 288 GB MI355X, its two
full-resolution levels run in z-slabs and, the tile being the periodic box itself, without halo recompute -- when that
is exact; --max-tile 256 / 0 restrict it to 256^3 tiles / disable it), with the input box and
the output boxes resident in HBM.  Arithmetic: float32-equivalent f16x3 split MFMA by default (`value`); the
strict float32 MFMA path is timed on the same box and reported under "strict_f32".  Weights are synthetic (seeded; the pretrained blob is not
available), which changes neither the FLOPs nor the bytes.  Rank 0 prints ONE JSON line.

N > 1: the box is sharded as bricks over the ranks (jax_nbody_emulator_with_dj_amd/sharding.py) -- z-slabs whenever a
slab is at least 48 planes deep: per step each rank exchanges with its two z neighbours 4 planes of raw input, 6 planes of
down_l0 output, 10 planes of down_l1 output and 4 planes of the level-0 skip connection over RCCL P2P (the two larger
transfers under compute, on a second stream) and otherwise works alone; total work is fixed (strong scaling).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0     # same guide: dense f16/bf16 MFMA
Z, OM = 0.5, 0.3


def cpu_baseline(threads, tile_budget_s=100.0):
    """the CPU restatement of the path (oracle/, float32, torch-CPU / oneDNN conv3d core under the oracle's
    wiring) on the GPU box's host cores, thread count stated, on bounded samples: BASELINE config 1 -- one
    (1,3,128,128,128) -> 32^3 forward, 4.317 TFLOP -- and, when config 1 predicts that it fits the budget, one 224^3 ->
    128^3 sub-box (35.86 TFLOP), the unit of config 3, extrapolated x 64 to the 512^3 box.  `value` is the config-3 rate
    (voxels of the box per second of 64 such sub-boxes); the config-1 rate is reported beside it."""
    import torch
    from oracle import layers as L, model as M, params as P, cosmology as C
    p = P.synthetic_params(seed=1234, mid_chan=64)
    Dz, vf = float(C.growth_factor(Z, OM)), float(C.vel_norm(Z, OM))
    keep = torch.get_num_threads()
    torch.set_num_threads(threads)

    def run(n):
        x = np.random.default_rng(0).standard_normal((1, 3, n, n, n)).astype(np.float32)
        t = time.perf_counter()
        with L.backend('torch'):
            d, v = M.forward(p, x, OM, Dz, vf, dtype=np.float32)
        dt = time.perf_counter() - t
        assert np.all(np.isfinite(d)) and np.all(np.isfinite(v))
        return dt

    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=threads):
        t1 = run(128)
        predicted = t1 * 35.86 / 4.317
        t224 = run(224) if predicted <= tile_budget_s else None
    torch.set_num_threads(keep)
    per_tile = t224 if t224 is not None else predicted
    out = {"value": 512.0 ** 3 / (64 * per_tile), "unit": "voxels/s", "cores": threads, "kind": "port",
           "sample": "float32 oracle with the torch-CPU (oneDNN) conv3d core, %d threads: config 1, one (1,3,128^3)->32^3 forward, "
                     "%.1f s (4.317 TFLOP, %.0f voxels/s); %s; value = 512^3 / (64 x that)"
                     % (threads, t1, 32 ** 3 / t1,
                        "one 224^3->128^3 sub-box, %.1f s (35.86 TFLOP)" % t224 if t224 is not None else
                        "one 224^3->128^3 sub-box NOT run (predicted %.0f s > %.0f s budget): extrapolated from config 1 by FLOPs" % (predicted, tile_budget_s)),
           "config1": {"value": 32 ** 3 / t1, "unit": "voxels/s", "seconds": t1, "tflop": 4.317},
           "subbox_224": {"seconds": t224, "tflop": 35.86, "measured": t224 is not None,
                          "tflops": 35.86 / t224 if t224 else 4.317 / t1}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="box side (debug; the headline is 512)")
    ap.add_argument("--ndiv", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vel", action="store_true")
    ap.add_argument("--max-tile", type=int, default=512,
                    help="cap on the internal tile edge: sub-boxes are merged into the largest tile that fits the "
                         "card's free memory when that is exact (crop %% 8 == 0): 512 -> four tiles of 256x256x512, "
                         "256 -> eight of 256^3; 0 = run the caller's 64 sub-boxes of 224^3 one by one")
    ap.add_argument("--precision", default=os.environ.get("NBE_PRECISION", "f16x3"), choices=["f32", "f16x3", "f16"],
                    help="f16x3 (default): float32-equivalent split-f16 MFMA, 3 MFMAs per product, f32 accumulate, "
                         "whole-network error vs the float64 oracle equal to or below the strict path's; "
                         "f32: strict float32 MFMA")
    ap.add_argument("--no-strict", action="store_true", help="skip the strict-f32 reference pass")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-array (NumPy in / NumPy out) pass")
    ap.add_argument("--no-small-configs", action="store_true", help="skip BASELINE configs 1 and 2 (128^3 cases) beside the headline")
    ap.add_argument("--no-profile", action="store_true",
                    help="no per-kernel HIP events in the timed region (no roofline object): the tiles then replay from "
                         "captured hipGraphs, which profiling turns off -- for the graph A/B (NBE_GRAPH=0 / 1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from jax_nbody_emulator_with_dj_amd import StyleNBodyEmulatorVelCore, StyleNBodyEmulatorCore, cosmology
    from jax_nbody_emulator_with_dj_amd.engine import Engine
    from jax_nbody_emulator_with_dj_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed launch with that many ranks" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    # test rig: NBE_BENCH_ONE_GPU=1 runs all ranks on cuda:0 with gloo halos (RCCL refuses several ranks per device);
    # the numbers of such a run mean nothing, it only exercises the N > 1 code path on a one-GPU box
    one_gpu = os.environ.get("NBE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        os.environ.setdefault("NBE_MEM_FRACTION", "%.3f" % (0.8 / max(world, 1)))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    vel = not args.no_vel
    model = (StyleNBodyEmulatorVelCore if vel else StyleNBodyEmulatorCore)()
    params = model.init(1234)
    Dz = float(np.float32(cosmology.growth_factor(Z, OM)))
    vf = float(np.float32(cosmology.vel_norm(Z, OM)))
    N = args.size
    size, ndiv = (N, N, N), (args.ndiv,) * 3
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    sb = None
    if world == 1:
        data = torch.randn((3,) + size, device=dev, dtype=torch.float32, generator=gen)
    else:
        grid = sharding.rank_grid(world, ndiv, size)
        _, bshape = sharding.brick_extent(sharding.rank_coords(rank, grid), grid, size)
        data = torch.randn((3,) + bshape, device=dev, dtype=torch.float32, generator=gen)
    disp = torch.zeros_like(data)
    velo = torch.zeros_like(data) if vel else None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    phases = {}
    graph_info = {}

    def measure(precision, warmup, steps):
        """W untimed + K timed passes of the whole box with one arithmetic mode."""
        nonlocal sb
        eng = Engine(device=local_rank, compute_vel=vel, precision=precision)
        eng.load_params(params, premodulated=False)
        eng.set_cosmology(OM, Dz)
        eng.set_max_tile(args.max_tile)
        if world == 1:
            step = lambda: eng.process_box(data, size, ndiv, ((48, 48),) * 3, Dz, vf, out=(disp, velo))
            plan = "%s tiles per box" % (eng.plan_tiles(size, ndiv),)
        else:
            sb = sharding.ShardedBox(eng, size, ndiv, rank, world, comm_stream=torch.cuda.Stream(device=dev))
            step = lambda: sb.process(data, Dz, vf, disp, velo)
            plan = None
        # with --warmup 0 one priming pass still runs untimed: the first pass of a process plans the tiles and allocates
        # and zero-fills a ~200 GB workspace (seconds), which is set-up, not the hot path
        for _ in range(max(warmup, 1)):
            step()
        fence()
        if plan is None:                          # (after the first step: the ranks agree on brick mode there, sharding.py)
            if sb.zbricks:
                plan = "one z-slab brick %s per rank, four face exchanges with the z neighbours (sharding.py)" % (sb.bshape,)
            else:
                plan = "%s padded tiles per rank brick %s" % (eng.plan_tiles(sb.bshape, sb.nd_local, periodic_box=False), sb.bshape)
        eng.debug_phase_cycles()                  # timing-probe builds: reset the in-kernel phase counters
        eng.profile_reset()
        eng.profile_enable(not args.no_profile)
        if sb is not None:
            sb.wait_ms(); sb.trace = True          # how long the compute stream stands still for the face exchanges
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if sb is not None:
            sb.trace = False
            w = sb.wait_ms()
            wt = torch.tensor([w.get("down_l0 faces", 0.0), w.get("down_l1 faces", 0.0)], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(wt, op=dist.ReduceOp.MAX)
            graph_info["exchange_wait_ms_per_step"] = {"down_l0 faces (6 planes, under the interior of conv_l1)": float(wt[0]) / steps,
                                                       "down_l1 faces (10 planes)": float(wt[1]) / steps,
                                                       "note": "max over ranks of the time the compute stream stood still; the skip-connection "
                                                               "planes are waited for inside nbe_brick_finish, after levels 1-3"}
        eng.profile_enable(False)
        prof = [] if args.no_profile else eng.profile_read()
        replays = eng.query("graph_replays")
        ph = eng.debug_phase_cycles()
        if ph[8] > 0:                             # NBE_BUILD_DBG=1 only: where the dominant kernel's wave cycles go
            tot = sum(ph[:8])
            phases.update({k: round(v / tot, 4) for k, v in zip(
                ("prologue", "stage_a_compute", "stage_a_own_dma", "stage_a_barrier", "stage_b_compute", "stage_b_own_dma",
                 "stage_b_barrier", "epilogue"), ph)})
            phases["cycles_per_wave"] = round(tot / ph[8])
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ok = bool(torch.isfinite(disp).all().item()) and (velo is None or bool(torch.isfinite(velo).all().item()))
        eng.close()
        graph_info["replays"] = replays
        return dt, prof, ok, plan

    def traffic_key(precision, plan):
        """What a PMC traffic figure was measured on (besides the kernel sources): tools/pmc_traffic.py --bench-json"""
        return "process_box %d^3 ndiv %d vel=%s precision=%s plan=%s periodic=%s" % (
            N, args.ndiv, vel, precision, plan.split(" tiles")[0], os.environ.get("NBE_PERIODIC", "1"))

    def roofline(prof, precision, plan="", kernel=None):
        # dominant kernel (or the named one), from HIP events recorded on the engine's stream inside the timed region.
        # f16x3 issues three f16 MFMAs per float32 product: algorithmic FLOPs are priced against 1/3 of the
        # dense f16 MFMA peak (2.5 PFLOP/s).  conv_h3w (Winograd F(2,3) along z) issues four plane convolutions where
        # the direct form has six, i.e. two MFMAs per algorithmic product: `frac` stays algorithmic FLOPs against the
        # same 833 TFLOP/s (what a direct kernel could reach at most), `mfma_issue_frac` is the share of the matrix
        # pipe's peak that the MFMAs it actually issues take.
        cand = [e for e in prof if kernel is None or e["kernel"].startswith(kernel)]
        if not cand:
            return None
        dom = max(cand, key=lambda e: e["ms"])
        tot_ms = sum(e["ms"] for e in prof)
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        peak = {"f32": PEAK_F32_MFMA_TFLOPS, "f16x3": PEAK_F16_MFMA_TFLOPS / 3.0, "f16": PEAK_F16_MFMA_TFLOPS}[precision]
        # HBM-side bytes per launch of the dominant kernel cannot be read from inside the process: they come from the
        # committed rocprofv3 --pmc passes of this exact workload (profiles/traffic.json, written by tools/pmc_traffic.py:
        # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) and are reported only while that file was measured on the kernel
        # sources this run uses (source hash) and on this workload; otherwise null.
        traffic = None
        try:
            from jax_nbody_emulator_with_dj_amd import _lib
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            wl = traffic_key(precision, plan)
            sub = {"conv_h3g": "conv_h3g_kernel<false,", "conv_h3n": "conv_h3g_kernel<true,", "conv_h3<FLAT3": "conv_h3q_kernel",
                   "conv_h3w": "conv_h3w_kernel", "conv_mfma_g": "conv_mfma_kernel"}
            key = next((v for k, v in sub.items() if dom["kernel"].startswith(k)), None)
            if tj.get("build") == _lib.source_hash() and tj.get("workload") == wl and key and world == 1:
                # (a profile entry may cover several instantiations of a kernel -- conv_h3w_kernel<false> / <true>: the
                # launch-weighted mean over all that match)
                m = [v for k, v in tj["kernels"].items() if key in k]
                if m:
                    traffic = sum(v["traffic_bytes"] * v["launches"] for v in m) / sum(v["launches"] for v in m)
        except Exception:
            traffic = None
        wino = dom["kernel"].startswith("conv_h3w") or dom["kernel"].startswith("conv_h1w")   # (conv_h1w: the float16 model's form)
        mpp = {"f32": 1.0, "f16x3": 3.0, "f16": 1.0}[precision] * (2.0 / 3.0 if wino else 1.0)
        full = {"f32": PEAK_F32_MFMA_TFLOPS}.get(precision, PEAK_F16_MFMA_TFLOPS)
        return {"bound": "mfma", "kernel": dom["kernel"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic, "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                "launches": dom["launches"], "share_of_kernel_time": dom["ms"] / tot_ms if tot_ms else None,
                "mfma_per_product": mpp, "mfma_issue_frac": ach * mpp / full}

    dt, prof, ok, plan = measure(args.precision, args.warmup, args.steps)
    main_replays = graph_info.get("replays")
    main_waits = graph_info.get("exchange_wait_ms_per_step")

    def measure_host_path(steps):
        """The reference's call shape (subbox.py:139-219): host NumPy array in, host NumPy arrays out, through the public
        API -- create_emulator(...).process_box(box, z, Om).  Warm (first call excluded, as README.md:241)."""
        from jax_nbody_emulator_with_dj_amd import create_emulator, SubboxConfig, models
        os.environ["NBE_PRECISION"] = args.precision
        emu = create_emulator(premodulate=False, compute_vel=vel, load_params=False,
                              processor_config=SubboxConfig(size=size, ndiv=ndiv))
        emu.params = params
        emu.processor.params = params
        models.get_engine(emu.model, local_rank, args.precision).set_max_tile(args.max_tile)
        box = data.cpu().numpy()                       # pageable host memory, like np.random.randn(...) in README.md:84
        # two untimed calls: the first plans, allocates the workspace and pins the output arrays; a loop that rebinds
        # `res` keeps the previous pair of fields alive during the next call, so the steady state cycles through two
        # pinned pairs from the pool -- the second call pins the second pair (hipHostMalloc of 3.2 GB is not hot-path work)
        # the reference's default call: process_box(input_box, z, Om) -- show_progress=True, a tqdm bar (subbox.py:139-146)
        res = emu.process_box(box, Z, OM)
        res = emu.process_box(box, Z, OM)
        t0 = time.perf_counter()
        for _ in range(steps):
            res = emu.process_box(box, Z, OM)
        dth = time.perf_counter() - t0
        piped = models.get_engine(emu.model, local_rank, args.precision).query("host_pipe") == 1.0
        r0 = res[0] if vel else res
        okh = bool(np.isfinite(r0[:, ::64, ::8, ::8]).all())
        same = None
        if vel and world == 1:                         # the resident run wrote `disp`: same box, same arithmetic
            same = bool(np.array_equal(r0[:, 100], disp[:, 100].cpu().numpy()))
        del res
        models.release_engines()
        return {"value": float(N) ** 3 * steps / dth, "unit": "voxels/s", "ms_per_step": 1e3 * dth / steps, "steps": steps,
                "call": "emu.process_box(box, z, Om) with default arguments (show_progress=True)", "pipelined": piped,
                "input": "pageable NumPy float32", "output": "NumPy float32 (pinned pool)", "finite": okh,
                "equals_resident_result": same}

    host = None
    if world == 1 and not args.no_host_path:
        host = measure_host_path(max(1, min(args.steps, 3)))
    strict = None
    if args.precision != "f32" and world == 1 and not args.no_strict:
        strict = measure("f32", 1, 1)             # the strict-float32 MFMA path on the same box, for reference

    def measure_small_configs(reps=5):
        """BASELINE configs 1 and 2 on this card, resident tensors (parity-test cases; reported beside the headline):
        config 1 = StyleNBodyEmulatorVelCore.apply on one (1,3,128,128,128) sub-box -> 32^3;
        config 2 = process_box 128^3, ndiv (1,1,1), compute_vel=False (the single-tile displacement-only kernel path)."""
        out = {}
        x = torch.randn((3, 128, 128, 128), device=dev, dtype=torch.float32, generator=gen)
        for name, cv in (("config1", True), ("config2", False)):
            eng = Engine(device=local_rank, compute_vel=cv, precision=args.precision)
            eng.load_params((StyleNBodyEmulatorVelCore if cv else StyleNBodyEmulatorCore)().init(1234), premodulated=False)
            eng.set_cosmology(OM, Dz)
            if cv:
                step = lambda: eng.forward(x, Dz, vf)
                work, nvox, flop = "apply (1,3,128,128,128) -> 32^3, disp + vel", 32 ** 3, 4.317e12
            else:
                step = lambda: eng.process_box(x, (128,) * 3, (1, 1, 1), ((48, 48),) * 3, Dz, 0.0)
                work, nvox, flop = "process_box 128^3 ndiv (1,1,1) compute_vel=False", 128 ** 3, None
            for _ in range(3):
                step()
            fence()
            eng.profile_reset(); eng.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            fence()
            dts = (time.perf_counter() - t0) / reps
            eng.profile_enable(False)
            pr = eng.profile_read()
            rf = roofline(pr, args.precision) if pr else None
            out[name] = {"workload": work, "value": nvox / dts, "unit": "voxels/s", "ms_per_step": 1e3 * dts, "steps": reps,
                         "reference_accounting_tflops": (flop / dts / 1e12) if flop else None,
                         "roofline": None if rf is None else {k: rf[k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "launches", "share_of_kernel_time")},
                         "plan": (eng.query("slab"), eng.query("periodic_yx"), eng.query("periodic_z"))}
            eng.close()
        return out

    small = measure_small_configs() if (world == 1 and not args.no_small_configs) else None

    if rank == 0:
        vox = float(N) ** 3
        dtype = {"f32": "f32", "f16x3": "f32-equivalent: f16x3 split MFMA (3 f16 MFMAs per product, f32 accumulate)",
                 "f16": "f16 (f16 operands and activations, f32 accumulate; the reference's dtype=float16 rows)"}
        out = {
            "metric": "voxels/sec (disp+vel) on 512^3 box, ndiv=4" if vel else "voxels/sec (disp only)",
            "value": vox * args.steps / dt, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype[args.precision], "data": "synthetic",
            "config": {"workload": "process_box %d^3 ndiv=(%d,%d,%d) compute_vel=%s StyleNBodyEmulator%sCore, "
                                   "synthetic seeded weights, box resident in HBM" % (N, *ndiv, vel, "Vel" if vel else ""),
                       "parallelism": "1 GPU" if world == 1 else "bricks %s + %s p2p halo exchange" % (
                           sb.grid, "RCCL (nccl backend, device to device)" if sb.backend() == "nccl" else
                           "%s (host-staged: the one-card test rig, not a multi-GPU measurement)" % sb.backend()),
                       "internal_tiles": plan, "precision": args.precision,
                       "traffic_key": traffic_key(args.precision, plan),
                       "tiles_replayed_from_hipgraphs": main_replays,
                       "exchange_wait_ms_per_step": main_waits},
            "finite": ok,
        }
        if prof:
            out["roofline"] = roofline(prof, args.precision, plan)
            if out["roofline"]["kernel"].startswith("conv_h3w"):      # launches that fell back to the direct gauged kernel, if any
                rd = roofline(prof, args.precision, plan, kernel="conv_h3g")
                if rd is not None:
                    out["roofline_direct_kernel"] = rd
            out["kernels"] = [{"kernel": e["kernel"], "ms": round(e["ms"], 3), "launches": e["launches"],
                               "tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 else None}
                              for e in sorted(prof, key=lambda e: -e["ms"])]
        if phases:
            out["phase_cycles_debug_build"] = phases
        if host is not None:
            host["vs_resident"] = host["value"] / out["value"]
            out["host_path"] = host
        if strict is not None:
            sdt, sprof, sok, _ = strict
            out["strict_f32"] = {"value": vox / sdt, "unit": "voxels/s", "ms_per_step": 1e3 * sdt, "steps": 1,
                                 "dtype": "f32", "finite": sok, "roofline": roofline(sprof, "f32") if sprof else None}
        if small is not None:
            out["baseline_configs_1_2"] = small
        if world == 1 and not args.no_cpu_baseline:
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = min(avail, 16)        # the 1-GPU box's CPU share; more BLAS threads only oversubscribe
            out["cpu_baseline"] = cpu_baseline(threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
