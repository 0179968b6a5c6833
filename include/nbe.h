/* nbe.h -- C ABI of the MI355X-native N-body emulator engine (libnbe.so).
 *
 * The reference (oleg-savchenko/jax_nbody_emulator_with_dj) has no native/FFI boundary: its boundary is
 * the Python API.  Each entry point below names the reference interface it replaces (file:line relative
 * to the reference repository root); the Python shim in jax_nbody_emulator_with_dj_amd/ binds them with
 * ctypes.
 *
 * Conventions: every function returns 0 on success and a non-zero code on failure (1 = error, 2 = NBE_ERANGE, see
 * "Range" below); the message is available from nbe_last_error() (thread local).  No entry point aborts the process.  Tensors are float32, C-contiguous, channel-first
 * ((C, D, H, W)) exactly as the reference passes them.  Data pointers may be host OR device pointers
 * (detected with hipPointerGetAttributes); the caller owns them.  The library owns all device memory it
 * allocates.  One in-flight call per context; several contexts may coexist (one per GPU / stream).
 */
#ifndef NBE_H
#define NBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbe_ctx nbe_ctx;

/* progress callback of nbe_process_box: replaces the tqdm bar of src/jax_nbody_emulator/subbox.py:186-193 */
typedef void (*nbe_progress_cb)(int done, int total, void* user);   /* done / total = fraction of the box finished */

/* One convolution layer of the parameter tree {'params': {block: {layer: {...}}}}
 * (leaf shapes: tests/test_style_nbody_emulator_vel_core.py:408-419, style_layers_vel.py:55-75;
 *  premodulated leaves: nbody_emulator.py:256-260).  All pointers are HOST float32. */
typedef struct nbe_layer_desc {
    const char* block;          /* "conv_l00", "down_l0", ...                         */
    const char* layer;          /* "skip", "conv_0", "conv_1"                           */
    int cout, cin, k;           /* weight shape (cout, cin, k, k, k)                    */
    const float* weight;        /* style: raw weight; premodulated: normalised weight   */
    const float* bias;          /* (cout,)                                              */
    const float* style_weight;  /* (cin, 2)  -- style trees only                        */
    const float* style_bias;    /* (cin,)    -- style trees only                        */
    const float* dweight;       /* premodulated velocity trees only, same shape as weight */
} nbe_layer_desc;

enum { NBE_F32 = 0, NBE_F16 = 1 };

const char* nbe_last_error(void);
int nbe_version(void);

/* context = one GPU + one stream + weights + workspace.
 * replaces: the implicit JAX device/jit state created by SubboxProcessor.__init__ (subbox.py:106-137) */
int nbe_create(int device_id, nbe_ctx** out);
int nbe_destroy(nbe_ctx* ctx);
/* run on a caller-provided hipStream_t (e.g. torch's current stream) so that the caller's device work before
 * and after a call is ordered with it; NULL = the device's default (null) stream, as everywhere in HIP.
 * A new context runs on its own non-blocking stream (NOT ordered with the null stream); nbe_use_own_stream
 * returns to it.  Both synchronise the stream being left. */
int nbe_set_stream(nbe_ctx* ctx, void* hip_stream);
int nbe_use_own_stream(nbe_ctx* ctx);
int nbe_synchronize(nbe_ctx* ctx);

/* model hyper-parameters: StyleNBodyEmulatorVelCore(style_size=2, in_chan, out_chan, mid_chan, eps)
 * (style_nbody_emulator_vel_core.py:39-43); compute_vel selects the *VelCore / *Core twin
 * (nbody_emulator.py:324-339). */
int nbe_set_arch(nbe_ctx* ctx, int in_chan, int out_chan, int mid_chan, float eps, int compute_vel);

/* Arithmetic of the convolutions (call before loading weights).  The reference selects it through the dtype
 * of x (SubboxConfig.dtype, style_layers_vel.py:103-105); its float32 runs on TF32-class tensor cores.
 *   NBE_PREC_F32    strict float32 MFMA (default)
 *   NBE_PREC_F16X3  float32-equivalent: operands split into two f16 numbers, three f16 MFMAs per product,
 *                   float32 accumulation (22-bit operands; measured whole-network error equal to float32's)
 *   NBE_PREC_F16    plain float16 operands, one f16 MFMA per product, float32 accumulation: the arithmetic of
 *                   the reference's SubboxConfig.dtype = float16 configuration (its fastest rows, README.md:245-250) */
enum { NBE_PREC_F32 = 0, NBE_PREC_F16X3 = 1, NBE_PREC_F16 = 2 };
int nbe_set_precision(nbe_ctx* ctx, int precision);

/* Range of the f16-based modes (NBE_PREC_F16X3, NBE_PREC_F16).  Their operands are f16 numbers (|v| < 65504, full
 * precision above 6.1e-5), while the reference's float32 arithmetic (style_layers_vel.py:103-105) has float32's range.
 * The engine therefore shifts every call into f16's comfortable range: LeakyReLU is positively homogeneous and the
 * convolutions are linear, so f(s x; s b) = s f(x; b) for the network f with input x and biases b, exactly in floating
 * point for s = 2^k.  Per call k is chosen such that max(max|x| * Dz / 6, max|b|) * 2^k lies in [0.5, 1) (NBE_PREC_F16)
 * or in [32, 64) (NBE_PREC_F16X3: the Winograd-z kernel keeps the lo part of its transformed planes unscaled, a normal f16
 * number for every |value| >= 2^-9 of the input's scale there): a reduction over the input, the input scaled in the
 * gather, the biases scaled on the device, 2^-k applied in the head.  Valid inputs: any finite float32 box -- parity
 * with the float64 oracle is tested over 24 decades of input scale (tests/test_gpu_range.py).  What remains out of
 * range is a network whose activations grow beyond 65504 (F16) / 1023 (F16X3) times its largest input / bias (weights far
 * from the unit-norm filters the modulation produces); then an infinity or a NaN reaches the head, which flags it:
 *   - host arrays in/out: the call returns NBE_ERANGE (2) instead of the fields;
 *   - device pointers (asynchronous calls): nbe_check_finite() synchronises and returns NBE_ERANGE if any call since
 *     the last check produced a non-finite value from a finite input.  The Python shim calls it after every call and
 *     recomputes that call on a strict-float32 context (never a silent inf / NaN).
 * Non-finite INPUT values propagate to the outputs as in the reference and are not an error.
 * nbe_set_input_range(ctx, m): use m as max|x| instead of reducing over the input (ranks of a sharded box agree on one
 * value with an all-reduce so that every brick is computed with the same shift); m < 0 returns to the reduction. */
enum { NBE_ERANGE = 2 };
int nbe_check_finite(nbe_ctx* ctx);
int nbe_set_input_range(nbe_ctx* ctx, float absmax);

/* state of the context after the last call / plan: see the enum */
enum { NBE_Q_GAUGE_ACTIVE = 0,     /* 1: the loaded weights run the two-product (gauged) tangent kernels             */
       NBE_Q_SLAB = 1,             /* planes per z-slab of the last plan (0 = whole tensors)                          */
       NBE_Q_PERIODIC_YX = 2,      /* 1: the last plan runs periodic in y and x                                       */
       NBE_Q_PERIODIC_Z = 3,       /* 1: ... and in z                                                                 */
       NBE_Q_RANGE_SHIFT = 4,      /* k of the last call's range shift 2^k                                            */
       NBE_Q_WORKSPACE_BYTES = 5,
       NBE_Q_HOST_PIPE = 6,        /* 1: the last nbe_process_box call ran the pipelined host path (nbe_host_alloc)    */
       NBE_Q_GRAPH_REPLAYS = 7,    /* tiles replayed from a captured hipGraph so far (see below)                       */
       NBE_Q_PLAN_TILES = 8,       /* tiles per box of the last plan                                                    */
       NBE_Q_PLAN_SHORT_GB = 9 };  /* > 0: the last plan is NOT the largest exact merge (max_tile permitting) because its
                                      workspace did not fit: GB of device memory that were missing.  The engine also
                                      writes one line to stderr when that happens (NBE_QUIET=1 silences it): the 512^3
                                      box as one tile needs ~200 GB free beside the box and the fields; with less the
                                      planner takes two, four or eight tiles and runs 1.1 - 1.4 x slower.              */
/* hipGraph replay.  A tile of nbe_process_box / nbe_process_region with device pointers in and out enqueues a few
 * hundred launches (the reference's analogue is the jitted step, subbox.py:137).  The second time the identical tile is
 * requested -- same pointers, geometry, scalars, weights and modulation -- its schedule is captured (on the context's
 * own stream, fenced against the caller's with events), and from the third time on it is ONE hipGraphLaunch.  Results
 * are bit-identical to the eager schedule.  NBE_GRAPH=0 disables it; profiling, progress callbacks and the pipelined
 * host path run eagerly. */
int nbe_query(nbe_ctx* ctx, int what, double* out);

/* replaces model.apply's `params` argument for the Style* cores (README.md:155; subbox.py:224-233) */
int nbe_load_style_weights(nbe_ctx* ctx, const nbe_layer_desc* layers, int nlayers);
/* replaces `params` of the premodulated cores: output of modulate_emulator_parameters[_vel]
 * (nbody_emulator.py:150-187, :221-266) */
int nbe_load_premod_weights(nbe_ctx* ctx, const nbe_layer_desc* layers, int nlayers);

/* Style cores: (Om, Dz) -> style vector s = ((Om-0.3)*5, Dz-1) and the per-layer weight modulation,
 * demodulation and d/dDz (style_nbody_emulator_vel_core.py:126-128, style_layers_vel.py:62-105).
 * Runs the modulate + pack kernels.  No-op for premodulated weights.
 * With velocity (f32 / f16x3) the tangent of style_layers_vel.py:98-105, dy = W.dx + dW.x, is evaluated as
 * W.(dx + alpha x) + beta (W.x) using dW = W (.) (alpha[cin] + beta[cout]) of the style modulation (two contractions
 * per 3x3x3 layer instead of three; same result within rounding).  env NBE_GAUGE=0 at load time keeps the
 * three-product form; a style factor that is exactly zero at (Om, Dz) selects it for that cosmology.
 * f16x3 with velocity: the wide 3x3x3 layers run a Winograd F(2,3) transform along z (conv_h3w_kernel: four plane-wise
 * convolutions per two output planes instead of six, same float32 tolerances).  Its rounding depends on how a launch pairs
 * its planes, so fields of different tilings / slab plans / rank counts agree to float32 rounding, not bit for bit;
 * env NBE_WINO=0 (read per launch) selects the direct kernel, whose rounding is independent of the schedule. */
int nbe_set_cosmology(nbe_ctx* ctx, float Om, float Dz);

/* model.apply(params, x[None], Om, Dz, vel_fac) for ONE batch element
 * (style_nbody_emulator_vel_core.py:105-195 and the three sibling signatures, subbox.py:224-233).
 * x: (in_chan, D, H, W); disp / vel: (out_chan, D-96, H-96, W-96); vel may be NULL when compute_vel=0. */
int nbe_forward(nbe_ctx* ctx, const void* x, int D, int H, int W, float Dz, float vel_fac,
                void* disp, void* vel);

/* SubboxProcessor.process_box (subbox.py:139-219): periodic 48-voxel-halo crops of `box`
 * ((in_chan, size0, size1, size2)), forward, ASSIGNMENT of the un-padded result into disp / vel
 * ((in_chan, size...), out_dtype NBE_F32 or NBE_F16).  Dz, vel_fac: growth_factor / vel_norm scalars
 * (subbox.py:173-178).  pad must be 48 on every side (the model's receptive field, subbox.py:43). */
int nbe_process_box(nbe_ctx* ctx, const void* box, const int64_t size[3], const int ndiv[3], const int pad[6],
                    float Dz, float vel_fac, void* disp, void* vel, int out_dtype,
                    nbe_progress_cb cb, void* user);

/* Pinned host memory from a process-wide pool (hipHostMalloc; freed buffers are kept for reuse up to NBE_PINNED_POOL_GB,
 * default 16).  Host-array calls of nbe_process_box whose OUTPUT arrays come from here are pipelined when the box runs
 * as one periodic tile (the default plan of a 512^3 box on a free MI355X): the input goes up in z-chunks through pinned
 * staging buffers filled by host threads while the first slabs run, and every finished output slab is copied out on a
 * second stream under the kernels of the next -- the reference does gather -> H2D -> compute -> D2H -> paste serially
 * per sub-box (subbox.py:195-215).  Plain (pageable) host arrays work as before, un-overlapped.  The Python shim
 * returns NumPy arrays backed by this pool.  nbe_host_trim releases the pooled buffers. */
void* nbe_host_alloc(size_t bytes);
int nbe_host_free(void* p);
int nbe_host_trim(void);

/* The same loop over a sub-set of the sub-boxes of a REGION of a periodic box (multi-GPU sharding: each
 * rank owns a brick).  Sub-boxes tile [origin, origin+region) with `ndiv`; `order`
 * (nullable) lists the sub-box indices to run, in order; results go to an output array of spatial size
 * out_size at out_origin + anchor.  Outputs are NOT zero-initialised here.  No reference counterpart:
 * the reference loop (subbox.py:195-215) is serial on one device. */
int nbe_process_region(nbe_ctx* ctx, const void* box, const int64_t box_size[3], const int64_t origin[3],
                       const int64_t region[3], const int ndiv[3], const int* order, int norder,
                       float Dz, float vel_fac, void* disp, void* vel, int out_dtype,
                       const int64_t out_size[3], const int64_t out_origin[3]);

/* Brick mode of a sharded box (no reference counterpart: the reference's loop is serial on one device, subbox.py:195-215).
 * The ranks of a node cut the periodic box into slabs along z; a brick is periodic in y and x by itself.  What the network
 * needs from the z neighbours is EXCHANGED at the three places where it is smallest, instead of being recomputed from a
 * 48-plane halo of the raw input:
 *   which 0   4 planes of the raw input per side -- the level-0 encoder's reach beyond the brick ((C, 4, S1, S2) float32);
 *   which 1   6 planes of the down_l0 output per side -- what conv_l1 reads beyond the brick for the level-1 skip connection;
 *   which 2   10 planes of the down_l1 output per side -- what levels 2 and 3 read;
 *   which 3   4 planes of the level-0 skip connection (conv_l01's output) per side -- what the decoder's first block reads
 *             beyond the brick; needed last, it travels while levels 1-3 run.
 * nbe_brick_halo_bytes(ctx, brick_size, which) sizes one such face (device buffers, opaque 16-byte units for 1 - 3).
 *   nbe_brick_encode    haloed_brick = (C, b0 + 8, S1, S2): level-0 encoder on the brick's own planes; writes the first / last
 *                       6 down_l0 planes to send_lo / send_hi and the first / last 4 skip-connection planes to skip_send_*;
 *   nbe_brick_interior  the part of conv_l1 that needs the brick's own planes only -- it runs while the faces travel;
 *   nbe_brick_exchange  with the neighbours' faces (recv_lo = the z-minus neighbour's send_hi, recv_hi = the z-plus
 *                       neighbour's send_lo): the rest of conv_l1, the skip connection, down_l1; writes the first / last
 *                       10 down_l1 planes to send2_lo / send2_hi;
 *   nbe_brick_finish    with the neighbours' second faces and their skip-connection planes: levels 2-3, the decoders, the
 *                       brick's (C, b0, S1, S2) fields.  The skip-connection planes are read last: the stream waits for
 *                       skip_ready_event (recorded by the caller behind their transfer; NULL = they are there) only after
 *                       levels 1-3, so that transfer is hidden under them.
 * The four calls must follow each other on one context (any other call in between invalidates the brick and the next
 * brick call fails); all are asynchronous on the context's stream -- the caller orders the exchanges against it (events).
 * EVERY RANK MUST USE THE SAME RANGE SHIFT: call nbe_set_input_range with the box-wide max |x| first (the shim all-reduces
 * it).  Fields equal the single-device nbe_process_box of the whole box: bit for bit on the direct kernels (NBE_WINO=0) and
 * whenever slab starts pair the planes alike, else to float32 rounding (conv_h3w_kernel pairs planes from the first plane of
 * a launch).  b0 must be a multiple of 8, at least 48.  nbe_brick_plan returns the planes per z-slab the brick would run
 * with on the memory that is free now, or 0 when it does not fit (the caller then takes nbe_process_region). */
int64_t nbe_brick_halo_bytes(nbe_ctx* ctx, const int64_t brick_size[3], int which);
int nbe_brick_plan(nbe_ctx* ctx, const int64_t brick_size[3]);
int nbe_brick_encode(nbe_ctx* ctx, const void* haloed_brick, const int64_t brick_size[3], float Dz, float vel_fac,
                     void* send_lo, void* send_hi, void* skip_send_lo, void* skip_send_hi);
int nbe_brick_interior(nbe_ctx* ctx);
int nbe_brick_exchange(nbe_ctx* ctx, const void* recv_lo, const void* recv_hi, void* send2_lo, void* send2_hi);
int nbe_brick_finish(nbe_ctx* ctx, const void* recv2_lo, const void* recv2_hi, const void* skip_recv_lo, const void* skip_recv_hi,
                     void* skip_ready_event /* hipEvent_t or NULL */, float Dz, float vel_fac, void* disp, void* vel, int out_dtype);

/* Internal tiling.  When crop_size = size/ndiv is a multiple of 8 on every axis, all crop origins keep the
 * phase of the network's 2^3 stride lattice, so the per-voxel result does not depend on how the box is cut
 *  and neighbouring sub-boxes can be merged into larger tiles that recompute less halo
 * (17.1 MFLOP/voxel at 128^3 crops, 11.2 at 256^3).  nbe_plan_tiles returns the grid nbe_process_box will
 * with a cubic cap: per axis the largest merge with tile edge <= max_tile; unchanged when crop % 8 != 0.
 * nbe_set_max_tile(ctx, 0) keeps the caller's grid exactly; nbe_set_max_tile(ctx, 256) restricts to 256^3 tiles. */
int nbe_plan_tiles(const int64_t region[3], const int ndiv[3], int max_tile, int out_ndiv[3]);
int nbe_set_max_tile(nbe_ctx* ctx, int max_tile);
/* The grid a context will actually run (weights loaded): among all exact merges with tile edge <= its max_tile
 * (default 512, or env NBE_MAX_TILE) the one with the largest tile whose workspace fits the device memory free at
 * the time of the call, longest along the last axis on ties (512^3 / ndiv 4 on a 288 GB MI355X: four tiles of
 * 256 x 256 x 512).  Falls back to the caller's grid when merging is not exact or no weights are loaded.
 * periodic_box != 0: `region` is a whole periodic box (nbe_process_box); tiles that span it in y and x then run in
 * periodic-yx mode (no halo recompute in y and x at the two full-resolution levels, a smaller workspace). */
int nbe_plan_tiles_ctx(nbe_ctx* ctx, const int64_t region[3], const int ndiv[3], int periodic_box, int out_ndiv[3]);
/* Schedule of the two full-resolution levels of the U-Net inside a tile: whole tensors, or slabs of `slab` output
 * planes along z (even; the slab-sized tensors let a tile be as deep as the box: 512^3 runs as ONE tile).  Results are
 * identical.  -1 (default, or env NBE_SLAB): chosen with the tiling by the memory that is free; 0: never; S: always. */
int nbe_set_slab(nbe_ctx* ctx, int slab);
/* Periodic-yx mode (default on, env NBE_PERIODIC=0 off): a tile of nbe_process_box that spans the periodic box in y
 * and x supplies the 48 voxels of y/x context of the two full-resolution levels layer by layer (1-voxel wrap-around
 * halos) instead of padding the input by 48 and shrinking.  Same arithmetic per voxel, ~10 % fewer FLOPs at 512^3. */
int nbe_set_periodic(nbe_ctx* ctx, int on);

/* growth_factor / vel_norm (cosmology.py:34-40, :130-141) in double precision on the host. */
double nbe_growth_factor(double z, double Om);
double nbe_vel_norm(double z, double Om);

/* ---- test / measurement hooks (not part of the reference surface) ------------------------------ */

/* One layer through the production kernels, host NCDHW in / out.  kind: 0 conv3 (VALID 3x3x3),
 * 1 skip (1x1x1, centre-cropped by `crop`), 2 down (k2 s2), 3 up (k2, lhs_dilation 2).
 * flags: 1 = LeakyReLU, 2 = add residual (res/dres shaped like the output).  dx/dw/dy/dres may be NULL. */
int nbe_test_layer(nbe_ctx* ctx, int kind, int crop, int flags, const float* x, const float* dx, int cin,
                   int D, int H, int W, const float* w, const float* dw, const float* bias, int cout,
                   const float* res, const float* dres, float* y, float* dy);
/* The gauged form of a 3x3x3 layer (style_layers_vel.py:98-105 with dW = W (.) (alpha[ci] + beta[co])):
 * dx is the input tangent in this layer's gauge, y = W.x + b, dy = W.dx + beta[o] * (W.x); flags: 1 = LeakyReLU.
 * f16x3 contexts run conv_h3w_kernel (Winograd F(2,3) along z) when the output has an even number of planes, Cin <= 128
 * and NBE_WINO is not 0, else conv_h3g_kernel. */
int nbe_test_layer_gauged(nbe_ctx* ctx, int flags, const float* x, const float* dx, int cin, int D, int H, int W,
                          const float* w, const float* beta, const float* bias, int cout, float* y, float* dy);
/* The same with flags bit 2: res / dres (shaped like the output) are added before the activation -- the float16 model's
 * conv_1 launches, whose Winograd-z form (conv_h3w_kernel<., ., F16>; Cin a multiple of 32) adds the block's skip there.
 * Other precisions run their direct gauged kernels when the residual flag is set. */
int nbe_test_layer_gauged_res(nbe_ctx* ctx, int flags, const float* x, const float* dx, int cin, int D, int H, int W,
                              const float* w, const float* beta, const float* bias, int cout, const float* res,
                              const float* dres, float* y, float* dy);
/* modulation kernel alone: OIDHW weight -> (w_n, dw_tot) */
int nbe_test_modulate(nbe_ctx* ctx, const float* weight, const float* style_weight, const float* style_bias,
                      int cout, int cin, int k, float s0, float s1, float eps, int first_layer,
                      float* w_n, float* dw_tot);

/* Branch probe -- test instrumentation for the kink-aware parity checks (tests/kink.py).
 * The tangent of LeakyReLUVel jumps by a factor 100 at zero (layers_vel.py:184-185), so two float evaluations of the
 * velocity differ wherever a pre-activation is zero to within rounding.  Armed with a block of nout^3 output voxels
 * (origin: coordinates in the output array of nbe_process_box / nbe_process_region / nbe_forward, multiples of 8; the
 * block must lie inside one tile of the plan), the next call records, for every LeakyReLU in the block's dependency cone
 * -- the 23 activation tensors of an (nout + 96)^3 input, core :105-195 -- which branch this library took.  The checker
 * then evaluates the float64 oracle on that cone WITH THESE BRANCHES and requires (a) agreement with the fields at the plain
 * tolerances on every voxel of the block and (b) that the branches differ from the oracle's own only where the oracle's
 * pre-activation is zero to within the tolerance of that tensor.  Works with every schedule (whole tensors, z-slabs,
 * periodic tiles, merged tiles) except brick mode.  hipGraph replay is off while a probe is armed.
 *   nbe_probe_slots    number of activation tensors recorded (23)
 *   nbe_probe_layout   slot i: "block/layer" of the convolution the activation follows, dims = {C, n, nw}: the bits of
 *                      the (C, n, n, n) cone tensor as (C, n, n, nw) 32-bit words, bit b of word w = voxel x = 32 w + b;
 *                      word_offset = its place in the buffer of nbe_probe_read
 *   nbe_probe_read     synchronises, checks that every voxel of every cone tensor was recorded exactly once, copies out */
int nbe_probe_begin(nbe_ctx* ctx, const int64_t origin[3], int nout);
int nbe_probe_slots(nbe_ctx* ctx);
int nbe_probe_layout(nbe_ctx* ctx, int slot, char* name, int name_cap, int dims[3], int64_t* word_offset);
int nbe_probe_read(nbe_ctx* ctx, void* words, int64_t nwords);
int nbe_probe_end(nbe_ctx* ctx);

/* per-kernel HIP-event timing on the engine's stream (bench.py roofline leg) */
int nbe_profile_enable(nbe_ctx* ctx, int on);
int nbe_profile_reset(nbe_ctx* ctx);
int nbe_profile_count(nbe_ctx* ctx);
/* entry i: kernel name, summed device ms, launches, algorithmic FLOPs (2*MAC of valid outputs) */
int nbe_profile_entry(nbe_ctx* ctx, int i, char* name, int name_cap, double* ms, int64_t* launches, double* flops);
/* bytes of device workspace currently held */
int64_t nbe_workspace_bytes(nbe_ctx* ctx);
/* timing-probe builds (-DNBE_DBG=1) only: s_memtime cycle totals per phase of the f16x3 3x3x3 kernel, summed over
 * waves since the last call: [0] prologue; [1]/[4] compute of the two stages of a (chunk, dz) group, [2]/[5] wait for
 * the wave's own DMA, [3]/[6] wait at the barrier; [7] epilogue; [8] number of waves.  Production builds return zeros. */
int nbe_debug_phase_cycles(nbe_ctx* ctx, double* out16);

#ifdef __cplusplus
}
#endif
#endif /* NBE_H */
