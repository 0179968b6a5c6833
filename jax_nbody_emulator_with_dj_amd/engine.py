"""Python handle on one libnbe context (one GPU, one stream, one set of weights)."""

import ctypes as C

import numpy as np

from . import _lib
from ._lib import LayerDesc, NBEError, NBERangeError, check   # noqa: F401  (re-exported)

try:  # torch is optional plumbing: device buffers + streams for resident / multi-GPU runs
    import torch
except Exception:  # pragma: no cover
    torch = None

BLOCKS = ('conv_l00', 'conv_l01', 'down_l0', 'conv_l1', 'down_l1', 'conv_l2', 'down_l2', 'conv_c',
          'up_r2', 'conv_r2', 'up_r1', 'conv_r1', 'up_r0', 'conv_r00', 'conv_r01')


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


try:
    from xxhash import xxh3_64_intdigest as _digest          # ~10 GB/s: 13 MB of weights in about a millisecond
except Exception:  # pragma: no cover
    from zlib import crc32 as _digest


def _leaf_digest(a):
    a = np.asarray(a)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    return (a.shape, a.dtype.str, _digest(memoryview(a).cast('B')))


def params_fingerprint(params):
    """Content fingerprint of a parameter tree (shape, dtype and a 64-bit checksum of every leaf).  Callers re-assign
    `.params` after construction (reference tests/test_nbody_emulator.py:807,832) and NumPy leaves are mutable, so the
    weights on the GPU are checked against the tree on every call; identity (`id`) is not enough -- an array edited
    in place keeps its id, and a freed tree's ids are recycled.  The reference re-reads `params` on every call."""
    tree = params['params'] if 'params' in params else params
    fp = []
    for b in sorted(tree):
        for l in sorted(tree[b]):
            for k in sorted(tree[b][l]):
                fp.append((b, l, k) + _leaf_digest(tree[b][l][k]))
    return tuple(fp)


class Engine:
    PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 2}

    def __init__(self, device=0, in_chan=3, out_chan=3, mid_chan=64, eps=1e-8, compute_vel=True, precision=None):
        """precision: "f16x3" (float32-equivalent split-f16 MFMA: three f16 MFMAs per product, float32
        accumulation; whole-network error vs float64 at or below the strict path's) or "f32" (strict float32
        MFMA) or "f16" (plain float16 operands, float32 accumulation: the reference's dtype=float16 arithmetic).
        Default: the environment variable NBE_PRECISION, else "f16x3"."""
        import os
        self._l = _lib.lib()
        if precision is None:
            precision = os.environ.get("NBE_PRECISION", "f16x3")
        if precision not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        h = C.c_void_p()
        check(self._l.nbe_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.precision = precision
        self.in_chan, self.out_chan, self.mid_chan = int(in_chan), int(out_chan), int(mid_chan)
        self.eps, self.compute_vel = float(eps), bool(compute_vel)
        check(self._l.nbe_set_arch(self._h, self.in_chan, self.out_chan, self.mid_chan, self.eps,
                                   1 if self.compute_vel else 0))
        check(self._l.nbe_set_precision(self._h, self.PRECISIONS[precision]))
        self._keep = None
        self.loaded = None          # fingerprint of the loaded tree
        self.premodulated = False

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._l.nbe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------
    def load_params(self, params, premodulated):
        if params is None:
            raise ValueError("No parameters loaded. Use load_params=True in create_emulator.")
        tree = params['params'] if 'params' in params else params
        descs, keep = [], []
        for b in BLOCKS:
            if b not in tree:
                raise NBEError("parameter tree is missing block %s" % b)
            for l, lp in tree[b].items():
                w = _f32(lp['weight'])
                bias = _f32(lp['bias'])
                d = LayerDesc()
                d.block, d.layer = b.encode(), l.encode()
                d.cout, d.cin, d.k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
                d.weight, d.bias = w.ctypes.data, bias.ctypes.data
                keep += [w, bias]
                if premodulated:
                    if 'style_weight' in lp:
                        raise NBEError("%s/%s holds style parameters but a premodulated model was requested" % (b, l))
                    if self.compute_vel:
                        if 'dweight' not in lp:
                            raise NBEError("%s/%s: premodulated velocity parameters need 'dweight'" % (b, l))
                        dw = _f32(lp['dweight'])
                        d.dweight = dw.ctypes.data
                        keep.append(dw)
                else:
                    if 'style_weight' not in lp:
                        raise NBEError("%s/%s has no style parameters; pass premodulated weights to the "
                                       "NBodyEmulator[Vel]Core models instead" % (b, l))
                    sw, sb = _f32(lp['style_weight']), _f32(lp['style_bias'])
                    d.style_weight, d.style_bias = sw.ctypes.data, sb.ctypes.data
                    keep += [sw, sb]
                descs.append(d)
        arr = (LayerDesc * len(descs))(*descs)
        fn = self._l.nbe_load_premod_weights if premodulated else self._l.nbe_load_style_weights
        check(fn(self._h, arr, len(descs)))
        self._keep = keep                # the library copies the weights; the float32 copies live as long as the load
        self.loaded = params_fingerprint(params)
        self.premodulated = bool(premodulated)

    def invalidate(self):
        """Forget what is loaded: the next call re-uploads the parameter tree."""
        self.loaded = None

    def ensure_params(self, params, premodulated):
        if params is None:
            raise ValueError("No parameters loaded. Use load_params=True in create_emulator.")
        if self.loaded != params_fingerprint(params) or self.premodulated != bool(premodulated):
            self.load_params(params, premodulated)

    def set_cosmology(self, Om, Dz):
        check(self._l.nbe_set_cosmology(self._h, float(Om), float(Dz)))

    def set_stream(self, stream_ptr):
        """Run on the hipStream_t `stream_ptr` (0 / None = the device's null stream)."""
        check(self._l.nbe_set_stream(self._h, C.c_void_p(stream_ptr) if stream_ptr else None))
        self._stream_ptr = int(stream_ptr or 0)

    def use_own_stream(self):
        check(self._l.nbe_use_own_stream(self._h))
        self._stream_ptr = None

    def set_max_tile(self, max_tile):
        """0 = run exactly the caller's sub-box grid; N = merge sub-boxes into tiles of edge <= N."""
        check(self._l.nbe_set_max_tile(self._h, int(max_tile)))
        self.max_tile = int(max_tile)

    def set_slab(self, slab):
        """-1: the engine chooses whole tensors or z-slabs by memory; 0: whole tensors; S (even): slabs of S planes."""
        check(self._l.nbe_set_slab(self._h, int(slab)))

    def set_periodic(self, on):
        """Periodic-yx mode for tiles that span the periodic box in y and x (default on)."""
        check(self._l.nbe_set_periodic(self._h, 1 if on else 0))

    def plan_tiles(self, region, ndiv, periodic_box=True):
        """The sub-box grid the engine will actually run for `region` cut by `ndiv` (nbe_plan_tiles_ctx): the
        largest exact merge whose workspace fits the free device memory.  periodic_box: `region` is a whole periodic
        box (process_box) rather than a brick of one (process_region)."""
        out = (C.c_int * 3)()
        check(self._l.nbe_plan_tiles_ctx(self._h, (C.c_int64 * 3)(*[int(r) for r in region]),
                                         (C.c_int * 3)(*[int(n) for n in ndiv]), 1 if periodic_box else 0, out))
        return tuple(out)

    def synchronize(self):
        check(self._l.nbe_synchronize(self._h))

    QUERY = {"gauge_active": 0, "slab": 1, "periodic_yx": 2, "periodic_z": 3, "range_shift": 4, "workspace_bytes": 5,
             "host_pipe": 6, "graph_replays": 7, "plan_tiles": 8, "plan_short_gb": 9}

    def query(self, what):
        """State of the context after the last call / plan (include/nbe.h, nbe_query)."""
        out = C.c_double()
        check(self._l.nbe_query(self._h, self.QUERY[what], C.byref(out)))
        return out.value

    def check_finite(self):
        """Synchronise and raise NBERangeError if a call since the last check produced non-finite values from a finite
        input (f16-based arithmetic, include/nbe.h "Range").  Host-array calls check themselves."""
        check(self._l.nbe_check_finite(self._h))

    def set_input_range(self, absmax):
        """max |input| of the following calls (None: the engine reduces over the input itself).  Ranks of a sharded box
        agree on one value so that every brick is computed with the same range shift."""
        check(self._l.nbe_set_input_range(self._h, -1.0 if absmax is None else float(absmax)))

    def _follow_torch_stream(self, t=None):
        """CUDA tensors in/out: enqueue on torch's current stream OF THIS ENGINE'S DEVICE so that torch ops before and
        after this call are ordered with the kernels (device pointers make the C calls asynchronous)."""
        if t is not None and (t.device.index or 0) != self.device:
            raise NBEError("tensor lives on cuda:%d but this engine was created for cuda:%d"
                           % (t.device.index or 0, self.device))
        sp = int(torch.cuda.current_stream(self.device).cuda_stream)      # 0 = torch's default = the null stream
        if getattr(self, '_stream_ptr', None) != sp:
            self.set_stream(sp)

    # ---- compute ------------------------------------------------------------------------------
    def forward(self, x, Dz, vel_fac=0.0):
        """x: (C, D, H, W) float32 numpy array or CUDA torch tensor.  Returns disp or (disp, vel)."""
        Cc, D, H, W = x.shape
        if Cc != self.in_chan:
            raise NBEError("input has %d channels, model expects %d" % (Cc, self.in_chan))
        oshape = (self.out_chan, D - 96, H - 96, W - 96)
        if min(oshape) <= 0:
            raise NBEError("input %s is smaller than the receptive field (needs > 96 per axis)" % (tuple(x.shape),))
        if _is_torch(x):
            self._follow_torch_stream(x)
            x = x.contiguous().float()
            disp = torch.empty(oshape, dtype=torch.float32, device=x.device)
            vel = torch.empty(oshape, dtype=torch.float32, device=x.device) if self.compute_vel else None
        else:
            x = _f32(x)
            disp = np.empty(oshape, np.float32)
            vel = np.empty(oshape, np.float32) if self.compute_vel else None
        check(self._l.nbe_forward(self._h, _ptr(x), D, H, W, float(Dz), float(vel_fac), _ptr(disp), _ptr(vel)))
        if _is_torch(x):
            self.check_finite()
        return (disp, vel) if self.compute_vel else disp

    def process_box(self, box, size, ndiv, padding, Dz, vel_fac=0.0, out_dtype=np.float32, progress=None,
                    out=None, check_finite=True):
        size = tuple(int(s) for s in size)
        if tuple(box.shape) != (self.in_chan,) + size:
            raise NBEError("input_box shape %s does not match (in_chan,)+size = %s" % (tuple(box.shape), (self.in_chan,) + size))
        half = np.dtype(out_dtype) == np.float16
        if not half and np.dtype(out_dtype) != np.float32:
            raise NBEError("output dtype %s unsupported (float32 / float16)" % np.dtype(out_dtype))
        oshape = (self.out_chan,) + size
        if _is_torch(box):
            self._follow_torch_stream(box)
            box = box.contiguous().float()
            tdt = torch.float16 if half else torch.float32
            if out is not None:
                disp, vel = out
            else:
                disp = torch.zeros(oshape, dtype=tdt, device=box.device)
                vel = torch.zeros(oshape, dtype=tdt, device=box.device) if self.compute_vel else None
        else:
            box = _f32(box)
            # pinned (pooled) outputs: the library fills every voxel -- computed ones by D2H, the rest with the zeros of
            # subbox.py:168-170 -- and copies finished slabs out while the next ones are computed
            disp = _lib.pinned_empty(oshape, np.float16 if half else np.float32)
            vel = _lib.pinned_empty(oshape, disp.dtype) if self.compute_vel else None
        sz = (C.c_int64 * 3)(*size)
        nd = (C.c_int * 3)(*[int(n) for n in ndiv])
        pd = (C.c_int * 6)(*[int(p) for pp in padding for p in pp])
        cb = _lib.PROGRESS_CB(progress) if progress is not None else C.cast(None, _lib.PROGRESS_CB)
        check(self._l.nbe_process_box(self._h, _ptr(box), sz, nd, pd, float(Dz), float(vel_fac), _ptr(disp),
                                      _ptr(vel), 1 if half else 0, cb, None))
        if _is_torch(box) and check_finite:     # device tensors: the call is asynchronous; check_finite=False defers it
            self.check_finite()
        return (disp, vel) if self.compute_vel else disp

    def process_region(self, box, origin, region, ndiv, Dz, vel_fac, disp, vel, out_origin=(0, 0, 0), order=None):
        """Run (a subset of) the sub-boxes tiling `region` of the periodic array `box` (CUDA tensors);
        results land in disp / vel (CUDA tensors) at out_origin + anchor.  Asynchronous on the engine's stream
        (call check_finite() when the step is complete)."""
        self._follow_torch_stream(box)
        i64 = lambda t: (C.c_int64 * 3)(*[int(v) for v in t])
        half = disp.element_size() == 2
        nd = (C.c_int * 3)(*[int(n) for n in ndiv])
        if order is not None:
            oarr = (C.c_int * len(order))(*[int(o) for o in order])
            if len(order) == 0:
                return
        else:
            oarr = None
        check(self._l.nbe_process_region(self._h, _ptr(box), i64(box.shape[1:]), i64(origin), i64(region), nd,
                                         oarr, 0 if order is None else len(order), float(Dz), float(vel_fac),
                                         _ptr(disp), _ptr(vel), 1 if half else 0, i64(disp.shape[1:]), i64(out_origin)))

    # ---- brick mode (sharded box, z-slabs with one activation exchange per box) -------------------
    RAW_HALO = 4            # planes of raw input a brick needs from either z neighbour (include/nbe.h, "Brick mode")

    def brick_halo_bytes(self, bshape, which=1):
        """Bytes of one exchanged face: which = 0 raw input (4 planes), 1 down_l0 output (6 planes), 2 down_l1 output (10),
        3 level-0 skip connection (4 planes)."""
        n = int(self._l.nbe_brick_halo_bytes(self._h, (C.c_int64 * 3)(*[int(v) for v in bshape]), int(which)))
        if n < 0:
            raise NBEError("nbe_brick_halo_bytes failed")
        return n

    def brick_plan(self, bshape):
        """Planes per z-slab the brick would run with on the memory that is free now; 0 = it does not fit."""
        return int(self._l.nbe_brick_plan(self._h, (C.c_int64 * 3)(*[int(v) for v in bshape])))

    def _check_faces(self, bshape, which, *faces):
        n = self.brick_halo_bytes(bshape, which)
        for t in faces:
            if not t.is_cuda or t.numel() * t.element_size() < n:
                raise NBEError("brick exchange buffers must be CUDA tensors of at least %d bytes" % n)

    def brick_encode(self, haloed, bshape, Dz, vel_fac, send_lo, send_hi, skip_send_lo, skip_send_hi):
        """haloed: CUDA tensor (C, b0 + 8, S1, S2); send_*: CUDA uint8 tensors of brick_halo_bytes(bshape, 1), skip_send_*: of
        brick_halo_bytes(bshape, 3)."""
        self._follow_torch_stream(haloed)
        b = tuple(int(v) for v in bshape)
        self._brick = b
        # the kernels read exactly this much: check before anything is launched
        want = (self.in_chan, b[0] + 2 * self.RAW_HALO, b[1], b[2])
        if tuple(haloed.shape) != want or haloed.dtype != torch.float32 or not haloed.is_contiguous():
            raise NBEError("haloed brick must be a contiguous float32 (C, b0 + 8, S1, S2) = %s tensor; got %s %s"
                           % (want, tuple(haloed.shape), haloed.dtype))
        self._check_faces(b, 1, send_lo, send_hi)
        self._check_faces(b, 3, skip_send_lo, skip_send_hi)
        check(self._l.nbe_brick_encode(self._h, _ptr(haloed), (C.c_int64 * 3)(*b), float(Dz), float(vel_fac),
                                       _ptr(send_lo), _ptr(send_hi), _ptr(skip_send_lo), _ptr(skip_send_hi)))

    def brick_interior(self):
        check(self._l.nbe_brick_interior(self._h))

    def brick_exchange(self, recv_lo, recv_hi, send2_lo, send2_hi):
        self._check_faces(self._brick, 1, recv_lo, recv_hi)
        self._check_faces(self._brick, 2, send2_lo, send2_hi)
        check(self._l.nbe_brick_exchange(self._h, _ptr(recv_lo), _ptr(recv_hi), _ptr(send2_lo), _ptr(send2_hi)))

    def brick_finish(self, recv2_lo, recv2_hi, skip_recv_lo, skip_recv_hi, Dz, vel_fac, disp, vel, skip_ready=None):
        """skip_ready: a torch.cuda.Event recorded behind the transfer of the skip-connection planes (None: they are there);
        the engine's stream waits for it only where the decoder first reads them, after levels 1-3."""
        self._follow_torch_stream(disp)
        half = disp.element_size() == 2
        self._check_faces(self._brick, 2, recv2_lo, recv2_hi)
        self._check_faces(self._brick, 3, skip_recv_lo, skip_recv_hi)
        ev = C.c_void_p(skip_ready.cuda_event) if skip_ready is not None else None
        check(self._l.nbe_brick_finish(self._h, _ptr(recv2_lo), _ptr(recv2_hi), _ptr(skip_recv_lo), _ptr(skip_recv_hi), ev,
                                       float(Dz), float(vel_fac), _ptr(disp), _ptr(vel), 1 if half else 0))

    # ---- test hooks ---------------------------------------------------------------------------
    def test_layer(self, kind, x, w, bias, dx=None, dw=None, crop=0, act=False, res=None, dres=None):
        kinds = {'conv3': 0, 'skip': 1, 'down': 2, 'up': 3}
        k = kinds[kind]
        x = _f32(x); w = _f32(w); bias = _f32(bias)
        cin, D, H, W = x.shape
        cout = w.shape[0]
        if k == 0: osp = (D - 2, H - 2, W - 2)
        elif k == 1: osp = (D - 2 * crop, H - 2 * crop, W - 2 * crop)
        elif k == 2: osp = (D // 2, H // 2, W // 2)
        else: osp = (2 * D, 2 * H, 2 * W)
        vel = dw is not None
        dx = None if dx is None else _f32(dx)
        dw = None if dw is None else _f32(dw)
        res = None if res is None else _f32(res)
        dres = None if dres is None else _f32(dres)
        y = np.empty((cout,) + osp, np.float32)
        dy = np.empty_like(y) if vel else None
        flags = (1 if act else 0) | (2 if res is not None else 0)
        check(self._l.nbe_test_layer(self._h, k, int(crop), flags, _ptr(x), _ptr(dx), cin, D, H, W, _ptr(w), _ptr(dw),
                                     _ptr(bias), cout, _ptr(res), _ptr(dres), _ptr(y), _ptr(dy)))
        return (y, dy) if vel else y

    def test_layer_gauged(self, x, dx, w, beta, bias, act=False, res=None, dres=None):
        """3x3x3 layer in the gauged form: y = W.x + b, dy = W.dx + beta[o] * (W.x) (nbe_test_layer_gauged);
        res / dres: a residual added before the activation (nbe_test_layer_gauged_res)."""
        x, dx, w, beta, bias = _f32(x), _f32(dx), _f32(w), _f32(beta), _f32(bias)
        cin, D, H, W = x.shape
        cout = w.shape[0]
        y = np.empty((cout, D - 2, H - 2, W - 2), np.float32)
        dy = np.empty_like(y)
        if res is not None:
            res, dres = _f32(res), _f32(dres)
            check(self._l.nbe_test_layer_gauged_res(self._h, (1 if act else 0) | 2, _ptr(x), _ptr(dx), cin, D, H, W, _ptr(w),
                                                    _ptr(beta), _ptr(bias), cout, _ptr(res), _ptr(dres), _ptr(y), _ptr(dy)))
            return y, dy
        check(self._l.nbe_test_layer_gauged(self._h, 1 if act else 0, _ptr(x), _ptr(dx), cin, D, H, W, _ptr(w), _ptr(beta),
                                            _ptr(bias), cout, _ptr(y), _ptr(dy)))
        return y, dy

    def test_modulate(self, weight, style_weight, style_bias, s, first_layer, eps=1e-8, vel=True):
        weight, sw, sb = _f32(weight), _f32(style_weight), _f32(style_bias)
        cout, cin, k = weight.shape[:3]
        wn = np.empty_like(weight)
        dw = np.empty_like(weight) if vel else None
        check(self._l.nbe_test_modulate(self._h, _ptr(weight), _ptr(sw), _ptr(sb), cout, cin, k, float(s[0]),
                                        float(s[1]), float(eps), 1 if first_layer else 0, _ptr(wn), _ptr(dw)))
        return (wn, dw) if vel else wn

    # ---- profiling ----------------------------------------------------------------------------
    # ---- branch probe (test instrumentation, include/nbe.h) -----------------------------------------
    def probe_begin(self, origin, nout=8):
        """Arm the branch probe for the block of nout^3 output voxels at `origin` (output-array coordinates)."""
        o = (C.c_int64 * 3)(*[int(v) for v in origin])
        check(self._l.nbe_probe_begin(self._h, o, int(nout)))

    def probe_read(self):
        """{'block/layer': bool array (C, n, n, n)} -- True where the tangent took the identity branch."""
        out, total, lay = {}, 0, []
        for i in range(self._l.nbe_probe_slots(self._h)):
            name = C.create_string_buffer(64)
            dims, off = (C.c_int * 3)(), C.c_int64()
            check(self._l.nbe_probe_layout(self._h, i, name, 64, dims, C.byref(off)))
            lay.append((name.value.decode(), dims[0], dims[1], dims[2], off.value))
            total = max(total, off.value + dims[0] * dims[1] * dims[1] * dims[2])
        words = np.empty(total, dtype=np.uint32)
        check(self._l.nbe_probe_read(self._h, words.ctypes.data_as(C.c_void_p), total))
        for name, c, n, nw, off in lay:
            w = words[off:off + c * n * n * nw].reshape(c, n, n, nw)
            bits = np.unpackbits(w.view(np.uint8), axis=-1, bitorder='little')      # little-endian words: bit b of word w = x 32 w + b
            out[name] = bits[..., :n].astype(bool)
        return out

    def probe_end(self):
        check(self._l.nbe_probe_end(self._h))

    def profile_enable(self, on=True):
        check(self._l.nbe_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(self._l.nbe_profile_reset(self._h))

    def profile_read(self):
        n = self._l.nbe_profile_count(self._h)
        out = []
        for i in range(n):
            name = C.create_string_buffer(128)
            ms, la, fl = C.c_double(), C.c_int64(), C.c_double()
            check(self._l.nbe_profile_entry(self._h, i, name, 128, C.byref(ms), C.byref(la), C.byref(fl)))
            out.append({'kernel': name.value.decode(), 'ms': ms.value, 'launches': la.value, 'flops': fl.value})
        return out

    def workspace_bytes(self):
        return int(self._l.nbe_workspace_bytes(self._h))

    def debug_phase_cycles(self):
        """Timing-probe builds only (NBE_BUILD_DBG=1): cycle totals per phase of the f16x3 3x3x3 kernel."""
        out = (C.c_double * 16)()
        check(self._l.nbe_debug_phase_cycles(self._h, out))
        return list(out)
