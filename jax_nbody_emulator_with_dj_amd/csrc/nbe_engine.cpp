// Engine behind the C ABI (include/nbe.h): weights, workspace planning, the U-Net schedule
// (style_nbody_emulator_vel_core.py:105-195) and the sub-box loop (subbox.py:139-219).
// Host code only; every device operation is a launch from nbe_kernels.hip.

#include "../../include/nbe.h"
#include "nbe_kernels.h"

#include <sched.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace nbe;

static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// workspace: first-fit allocator over one device block; a dry run sizes it
// ------------------------------------------------------------------------------------------------
struct Arena {
    struct Blk { int64_t off, size; bool used; };
    std::vector<Blk> blks;
    int64_t high = 0;
    void reset() { blks.clear(); blks.push_back({0, INT64_MAX / 2, false}); high = 0; }
    int64_t alloc(int64_t bytes) {
        bytes = (bytes + 255) & ~int64_t(255);
        for (size_t i = 0; i < blks.size(); ++i) {
            if (!blks[i].used && blks[i].size >= bytes) {
                Blk rest{blks[i].off + bytes, blks[i].size - bytes, false};
                blks[i].size = bytes; blks[i].used = true;
                if (rest.size > 0) blks.insert(blks.begin() + i + 1, rest);
                if (blks[i].off + bytes > high) high = blks[i].off + bytes;
                return blks[i].off;
            }
        }
        return -1;
    }
    void release(int64_t off) {
        for (size_t i = 0; i < blks.size(); ++i) {
            if (blks[i].off == off && blks[i].used) {
                blks[i].used = false;
                if (i + 1 < blks.size() && !blks[i + 1].used) { blks[i].size += blks[i + 1].size; blks.erase(blks.begin() + i + 1); }
                if (i > 0 && !blks[i - 1].used) { blks[i - 1].size += blks[i].size; blks.erase(blks.begin() + i); }
                return;
            }
        }
    }
};

struct Layer {
    std::string block, layer;
    int cout = 0, cin = 0, k = 0, kind = 0;       // kind: 0 conv3, 1 skip, 2 down, 3 up
    bool first = false;                           // conv_l00/{conv_0,skip}: input linear in Dz
    float *weight = nullptr, *sw = nullptr, *sb = nullptr;   // raw style parameters (device)
    float *wn = nullptr, *dwn = nullptr;          // modulated OIDHW (device)
    PackedW pw;
    PackedW pwn;                                  // narrow (16-cout tile) packing of w for the gauged 3x3x3 kernel: cout <= 16
    // tangent gauge (style path, see conv_h3g_kernel): dw = w_n (.) (alpha[ci] + beta[co])
    float* bias0 = nullptr;                       // the bias as loaded (device, padded like pw.bias, which holds bias0 * act_scale)
    float *alpha = nullptr, *beta = nullptr;      // this layer's own factors (device; cin / cout entries, zero-padded)
    const float* gout = nullptr;                  // gauge of the output tensor = alpha of its 3x3x3 consumer (+ channel offset)
    const float* a_in = nullptr;                  // general kernels: gauge of the input tensor, folded into dw
    bool g6 = false;                              // 3x3x3 layer whose input arrives in its own gauge: two products, no dw
    // Skip fusion (conv_h3g_kernel): a block's conv_1 computes the block's 1x1x1 skip as extra groups on the block input.
    const Layer* fskip = nullptr;                 // conv_1: the block's skip layer, when the block can run fused
    const float* b_sub = nullptr;                 // skip: beta of the block's conv_1, folded into dW_s~ when fused
    float* bias_f = nullptr;                      // conv_1: (b_1 + b_s) * act_scale (device, padded like pw.bias)
    // float16 model: a fused skip exists in the Winograd-z kernel only, and a launch that has no Winograd-z form (an odd number
    // of planes, NBE_WINO=0) runs the block unfused -- so the skip keeps its tangent weights in both versions: dwn without
    // conv_1's beta (pw.dw, the skip's own launch) and dwn_f with it folded in (pw.ww, the fused stages)
    float* dwn_f = nullptr;
};

// pad > 0: the tensor carries a periodic halo of `pad` voxels in y and x around its interior (periodic-yx mode)
// org: index, in the frame of the tensor the oracle forms for this layer on the tile's padded input, of the interior
// voxel (0, 0, 0) -- only the branch probe reads it (whole tensors of a padded tile: all zero)
struct Tensor { Planes p; int64_t off = -1; int pad = 0; int org[3] = {0, 0, 0}; };

struct ProfEntry { std::string name; double ms = 0; int64_t launches = 0; double flops = 0; };

// Progress reports that do not stall the stream.  The reference's process_box shows a tqdm bar by default
// (subbox.py:139-146, :186-193), so the default call carries a callback: the schedule records an event where a unit of work
// ends (a decoder slab's results on their way to the host, a tile) and this thread calls the callback once the event has
// completed -- nothing on the enqueueing side waits for the GPU.
struct Progress {
    struct Item { hipEvent_t ev; int done, total; };
    nbe_progress_cb cb; void* user; int device;
    std::thread th; std::mutex mu; std::condition_variable cv; std::deque<Item> q; bool stop = false;
    Progress(nbe_progress_cb cb_, void* user_, int dev) : cb(cb_), user(user_), device(dev) {
        th = std::thread([this] {
            (void)hipSetDevice(device);
            for (;;) {
                Item it;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    it = q.front(); q.pop_front();
                }
                (void)hipEventSynchronize(it.ev);
                (void)hipEventDestroy(it.ev);
                cb(it.done, it.total, user);
            }
        });
    }
    void post(hipStream_t s, int done, int total) {              // "done of total" holds once everything enqueued on s so far has run
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return;
        (void)hipEventRecord(ev, s);
        { std::lock_guard<std::mutex> lk(mu); q.push_back({ev, done, total}); }
        cv.notify_one();
    }
    ~Progress() {                                               // reports what is queued, then joins
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_one();
        if (th.joinable()) th.join();
    }
};

struct nbe_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int in_chan = 3, out_chan = 3, mid = 64;
    float eps = 1e-8f;
    bool vel = true;
    bool have_weights = false, style = false, modulated = false;
    float mod_Om = NAN, mod_Dz = NAN;
    std::map<std::string, Layer> layers;
    // workspace
    Arena arena;
    char* ws = nullptr;
    int64_t ws_bytes = 0;
    bool dry = false;
    int slab = 0;                                 // z-slab schedule: planes per slab of the full-resolution levels (0 = whole tensors)
    int slab_forced = -1;                         // -1: chosen by memory; 0: never; S > 0: always S (nbe_set_slab, env NBE_SLAB)
    bool pyx = false;                             // current tile runs in periodic-yx mode (it spans the periodic box in y and x)
    bool pyx_allowed = true;                      // env NBE_PERIODIC=0 turns the mode off
    bool pz = false;                              // ... and the tile also spans the box in z (only with pyx)
    // Brick mode of the sharded box (nbe_brick_encode / _interior / _exchange / _finish): the tile is one rank's z-slab of
    // the periodic box, periodic in y and x; in z it runs like pz, except that what the levels read beyond the brick's own
    // planes comes from the neighbours (four exchanges between the calls, network_stream) instead of periodic wrap-around.
    bool zx = false;
    int phase = 0;                                // 0: whole schedule; 1 .. 4: the four brick calls (network_stream)
    struct BrickIO { void *send_lo = nullptr, *send_hi = nullptr; const void *recv_lo = nullptr, *recv_hi = nullptr;
                     void *skip_send_lo = nullptr, *skip_send_hi = nullptr; const void *skip_recv_lo = nullptr, *skip_recv_hi = nullptr;
                     hipEvent_t skip_ready = nullptr; } bio;
    struct StreamState {                          // what the next brick call resumes with (tensors in the arena, which is left alone in between)
        bool valid = false;
        int stage = 0;                            // the last phase that ran (1 encode, 2 interior, 3 edges)
        Tensor skip0, td, tin, t, h, y1, cat1, t2;
        int D = 0, H = 0, W = 0, S = 0;
        std::vector<Arena::Blk> blks; int64_t high = 0;
        float Dz = 0.f, vel_fac = 0.f, act_scale = 1.f; const char* ws = nullptr;   // what the later calls must be made with
    } sst;
    // progress inside a tile (z-slab schedule): tile k of n, reported in thousandths of a tile
    nbe_progress_cb prog_cb = nullptr; void* prog_user = nullptr; int prog_k = 0, prog_n = 1;
    Progress* prog = nullptr;                     // the reporter of the running call (process_region owns it)
    int max_tile = 512;                           // cap on the internal tile edge (output voxels); 0 = caller's grid as given
    int prec = PREC_F32;                          // arithmetic of the convolutions (nbe_set_precision)
    bool gauge = false;                           // the loaded network is wired for gauged tangents (style weights, velocity)
    bool gauge_active = false;                    // ... and the current modulation uses them (no style factor is zero)
    bool fuse = false;                            // ... and the blocks' skips run fused into their conv_1 (f16x3 only)
    bool novel_fuse = false;                      // displacement-only f16x3: the blocks are wired for conv_h3w_kernel<SKIP, NOVEL>
    int64_t bp_size[3] = {0, 0, 0}; int bp_slab = 0; int64_t bp_need = 0;   // the brick plan that nbe_brick_plan / nbe_brick_encode last made
    int plan_tiles = 0;                           // tiles per box of the last plan (nbe_query)
    double plan_short_gb = 0.0;                   // > 0: a larger exact merge existed but its workspace lacked this much memory
    int64_t plan_logged = 0;                      // the situation the last stderr line was about (one line per situation)
    int* gauge_flag = nullptr;                    // device flag of launch_style_alpha
    // Winograd-z form of the gauged 3x3x3 layers (conv_h3w_kernel): packed beside pw.w for every gauged wide layer;
    // wino_ok is cleared when a weight of the current modulation leaves the f16 range at the kernel's 2^14 scale
    int* wino_flag = nullptr; bool wino_ok = false;
    // Range shift of the f16-based arithmetic (include/nbe.h, "Range"): activations and biases of a call are multiplied
    // by act_scale = 2^k (exact), the head divides it out.  flags[0]: bit pattern of max |input| (launch_absmax),
    // flags[1]: a non-finite value was written by the head.
    float act_scale = 1.f;                        // 2^k of the current call
    float bias_scale = 1.f;                       // 2^k the device biases currently carry
    bool bias_dirty = true;                       // the scaled biases (pw.bias, bias_f) have to be rewritten (new weights)
    float bias_max = 0.f;                         // max |bias| over all layers (host, at load time)
    float preset_absmax = -1.f;                   // >= 0: max |input| supplied by the caller (nbe_set_input_range)
    bool input_finite = true;                     // the input of the current call had no NaN / infinity
    bool range_pending = false;                   // a call has run since the last nbe_check_finite
    unsigned* flags = nullptr;                    // device: [0] absmax bits, [1] non-finite output
    // device-resident boxes of process_box
    float* box_in = nullptr; int64_t box_in_bytes = 0;
    char* box_out = nullptr; int64_t box_out_bytes = 0;
    // Host-array calls of process_box (the reference's call shape, subbox.py:168-170, :195-215), pipelined: the input
    // box goes up in z-chunks through pinned staging buffers while the encoder slabs run, every finished output slab
    // comes down on a copy stream under the next slab's kernels (HostPipe, below)
    struct HostPipe {
        bool active = false, out_async = false;
        bool tiles = false;                       // several tiles: tile k+1's planes go up and tile k-1's results come down under tile k
        bool slabwise = false;                    // ... and the running tile gathers slab by slab as its planes land (z-slab schedule)
        int o1 = 0, o2 = 0;                       // y / x origin of the running tile's gather (tiles mode; the one-tile plan: -halo)
        const float* hbox = nullptr;              // caller's (C, S0, S1, S2) array
        bool in_pinned = false;
        int C = 0, S0 = 0, S1 = 0, S2 = 0, o0 = 0;    // o0: box plane of tile plane 0 (may be negative: periodic)
        std::vector<char> up;                     // box plane uploaded?
        int gz = 0;                               // tile planes [.., gz) have been gathered
        char *hdisp = nullptr, *hvel = nullptr;   // caller's output arrays (pinned)
        char *ddisp = nullptr, *dvel = nullptr;   // device staging of the outputs
        int esz = 4, O0 = 0, O1 = 0, O2 = 0;
        int nstage = 0;                           // chunks staged so far (ring position)
    } pipe;
    bool last_piped = false;                      // the last process_box / process_region call ran pipelined
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    static constexpr int NSTAGE = 3;
    char* stage_buf[NSTAGE] = {nullptr, nullptr, nullptr}; int64_t stage_bytes = 0;
    hipEvent_t stage_free[NSTAGE] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_up = nullptr, ev_down = nullptr;
    // hipGraph replay of a tile's schedule (run_tile): everything a tile enqueues -- ~300 launches for the 512^3 box as
    // one tile -- is captured the second time the same tile is asked for and replayed from then on
    struct GraphKey {
        const void *box, *disp, *velo, *ws;
        int geo[18]; float f[3]; int epoch, flags;
        bool operator<(const GraphKey& o) const { return memcmp(this, &o, sizeof *this) < 0; }
    };
    struct GraphVal { hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; int seen = 0; uint64_t used = 0; };
    std::map<GraphKey, GraphVal> graphs;
    uint64_t graph_clock = 0, graph_replays = 0;
    int epoch = 0;                                // bumped whenever weights, modulation or schedule switches change
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;
    // Branch probe (test instrumentation, include/nbe.h): which LeakyReLU branch every activation in the dependency
    // cone of a block of output voxels took
    struct Probe {
        bool on = false, tile = false;            // armed; the tile being run contains the block
        int p[3] = {0, 0, 0}, nout = 0;           // block origin (output array coordinates) and edge
        int o[3] = {0, 0, 0};                     // ... in the frame of the running tile's padded input (level 0)
        struct Slot { std::string name; int C, n, nw, level; int64_t off; };
        std::vector<Slot> slots;
        unsigned* bits = nullptr; int64_t words = 0;
        unsigned* count = nullptr;                // per slot: words written
        int zr[3] = {0, 0, 0};                    // set by the schedule around a block: planes [lo, hi) of its result exist, period (0: no wrap)
    } probe;
    // profiling
    bool prof = false;
    std::vector<ProfEntry> prof_entries;
    struct Pending { int entry; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
};

// ------------------------------------------------------------------------------------------------
// cosmology scalars (cosmology.py:24-40, :101-141), double precision, own 2F1 series
// ------------------------------------------------------------------------------------------------
static double hyp2f1_series(double a, double b, double c, double z) {
    double term = 1.0, sum = 1.0;
    for (int n = 0; n < 200000; ++n) {
        term *= (a + n) * (b + n) / ((c + n) * (n + 1.0)) * z;
        sum += term;
        if (std::fabs(term) < 1e-17 * std::fabs(sum)) break;
    }
    return sum;
}
static double hyp2f1(double a, double b, double c, double x) {
    if (x < 0) return std::pow(1.0 - x, -a) * hyp2f1_series(a, c - b, c, x / (x - 1.0));   // Pfaff
    return hyp2f1_series(a, b, c, x);
}
static const double A2 = 1.0, B2 = 1.0 / 3.0, C2 = 11.0 / 6.0;

extern "C" double nbe_growth_factor(double z, double Om) {
    const double a = 1.0 / (1.0 + z), OL = 1.0 - Om;
    return a * hyp2f1(A2, B2, C2, -OL * a * a * a / Om) / hyp2f1(A2, B2, C2, -OL / Om);
}
static double growth_rate(double z, double Om) {
    const double a = 1.0 / (1.0 + z), x = -(1.0 - Om) * a * a * a / Om;
    const double F = hyp2f1(A2, B2, C2, x);
    const double dF = (A2 * B2 / C2) * hyp2f1(A2 + 1, B2 + 1, C2 + 1, x);
    return 1.0 + 3.0 * x * dF / F;
}
extern "C" double nbe_vel_norm(double z, double Om) {
    const double H = 100.0 * std::sqrt(Om * std::pow(1.0 + z, 3) + (1.0 - Om));
    return nbe_growth_factor(z, Om) * growth_rate(z, Om) * H / (1.0 + z);
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
static int roundup(int v, int m) { return (v + m - 1) / m * m; }

// ---- host-side helpers of the pipelined host path -------------------------------------------------------------
static int host_threads() {
    static const int n = [] {
        if (const char* e = getenv("NBE_HOST_THREADS")) return std::max(1, atoi(e));
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t set;                                          // the cores this process may actually use
        if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min<unsigned>(hw ? hw : 64, (unsigned)CPU_COUNT(&set));
        return (int)std::max(1u, std::min(hw ? hw : 4u, 16u));
    }();
    return n;
}
template <typename F>
static void parallel_for(int nt, F&& fn) {                      // fn(i, nt) on nt threads (the caller runs share 0)
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back([&fn, i, nt] { fn(i, nt); });
    fn(0, nt);
    for (auto& t : th) t.join();
}
static void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    const int nt = bytes < (size_t(8) << 20) ? 1 : host_threads();
    parallel_for(nt, [&](int i, int n) {
        const size_t per = ((bytes + n - 1) / n + 4095) & ~size_t(4095), b = std::min(bytes, per * i), e = std::min(bytes, b + per);
        if (e > b) memcpy((char*)dst + b, (const char*)src + b, e - b);
    });
}
// bit pattern of max |x| over a host array (the host-side twin of launch_absmax)
static unsigned host_absmax_bits(const float* x, int64_t n) {
    const int nt = n < (1 << 22) ? 1 : host_threads();
    std::vector<unsigned> part(nt, 0u);
    parallel_for(nt, [&](int i, int k) {
        const int64_t per = (n + k - 1) / k, b = std::min(n, per * i), e = std::min(n, b + per);
        const unsigned* u = (const unsigned*)x;
        unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        int64_t j = b;
        for (; j + 4 <= e; j += 4) {
            m0 = std::max(m0, u[j] & 0x7fffffffu); m1 = std::max(m1, u[j + 1] & 0x7fffffffu);
            m2 = std::max(m2, u[j + 2] & 0x7fffffffu); m3 = std::max(m3, u[j + 3] & 0x7fffffffu);
        }
        for (; j < e; ++j) m0 = std::max(m0, u[j] & 0x7fffffffu);
        part[i] = std::max(std::max(m0, m1), std::max(m2, m3));
    });
    unsigned m = 0;
    for (unsigned v : part) m = std::max(m, v);
    return m;
}
static bool is_pinned_host_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// ---- pinned host memory pool (nbe_host_alloc / nbe_host_free) ---------------------------------------------------
static std::mutex g_pin_mu;
static std::multimap<size_t, void*> g_pin_free;                 // size -> buffer, ready for reuse
static std::map<void*, size_t> g_pin_live;                      // handed out
static size_t g_pin_free_bytes = 0;
static size_t pin_pool_cap() {
    static const size_t cap = (size_t)((getenv("NBE_PINNED_POOL_GB") ? atof(getenv("NBE_PINNED_POOL_GB")) : 16.0) * (1ull << 30));
    return cap;
}
// every consumer reads whole 16-channel chunks; PREC_F16 stores 8 channels per plane, the others 4 (or hi+lo of 8)
static int planes_for(int C, int prec) { return roundup(C, 16) / (prec == PREC_F16 ? 8 : 4); }

static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice;
}

static Planes ws_planes(nbe_ctx* c, int G, int D, int H, int W, int64_t* off_out) {
    Planes p;
    p.G = G; p.D = D; p.H = H; p.W = W;
    p.pstride = (p.vox() + 63) & ~int64_t(63);
    const int64_t one = (int64_t)G * p.pstride * 16;
    // slack: the conv kernels stream whole row segments and may read up to 2*H*W + 2*W + ~600 voxels past the
    // end of a plane for flat positions whose outputs are discarded (the last plane must not run off the arena)
    const int64_t slack = ((int64_t)2 * H * W + 2 * W + 1024) * 16;
    const int64_t off = c->arena.alloc(one * (c->vel ? 2 : 1) + slack);
    *off_out = off;
    if (!c->dry) {
        p.x = (float*)(c->ws + off);
        p.dx = c->vel ? (float*)(c->ws + off + one) : nullptr;
    }
    return p;
}

static Tensor talloc(nbe_ctx* c, int C, int D, int H, int W) {
    Tensor t;
    t.p = ws_planes(c, planes_for(C, c->prec), D, H, W, &t.off);
    // Channel planes beyond C (C not a multiple of 16: narrow test models) are read by the consumer against zero
    // weights but never written by the producer: they must hold finite values whatever an earlier call -- another
    // shape, a NaN in its input, an overflow -- left at this place of the arena.  No such planes at production width.
    const int gw = c->prec == PREC_F16 ? (C + 7) / 8 : c->prec == PREC_F16X3 ? 2 * ((C + 7) / 8) : (C + 3) / 4;
    if (!c->dry && t.off >= 0 && gw < t.p.G) {
        const size_t off = (size_t)gw * t.p.pstride * 4, bytes = (size_t)(t.p.G - gw) * t.p.pstride * 16;
        launch_zero(t.p.x + off, (int64_t)bytes, c->stream);
        if (t.p.dx) launch_zero(t.p.dx + off, (int64_t)bytes, c->stream);
    }
    return t;
}
// interior Hi x Wi plus a y/x halo of `pad`
static Tensor tallocp(nbe_ctx* c, int C, int D, int Hi, int Wi, int pad) {
    Tensor t = talloc(c, C, D, Hi + 2 * pad, Wi + 2 * pad);
    t.pad = pad;
    return t;
}
// the interior as output planes: same strides, origin moved by (pad, pad)
static Planes inner(const Tensor& t) {
    Planes p = t.p;
    const int64_t sh = (int64_t)t.pad * (t.p.W + 1) * 4;
    if (p.x) p.x += sh;
    if (p.dx) p.dx += sh;
    return p;
}
static void fill_halo(nbe_ctx* c, const Tensor& t) {
    if (t.pad > 0 && !c->dry) launch_fill_yx(t.p, t.pad, c->vel, c->stream);
}
static void tfree(nbe_ctx* c, Tensor& t) { if (t.off >= 0) c->arena.release(t.off); t.off = -1; }
// planes [z0, z0 + nz) of every channel plane of t as a tensor of its own (not owning: pstride, H, W unchanged)
static Tensor zview(const Tensor& t, int z0, int nz) {
    Tensor v = t;
    v.off = -1;
    const int64_t sh = (int64_t)z0 * t.p.H * t.p.W * 4;          // floats: one voxel of a plane is 16 bytes
    if (v.p.x) v.p.x += sh;
    if (v.p.dx) v.p.dx += sh;
    v.p.D = nz;
    v.org[0] += z0;
    return v;
}
// frame bookkeeping of the branch probe: a tensor produced from x by `nconv` 3x3x3 layers (VALID: same origin; periodic
// in y and x: the interior's origin moves one voxel out per layer)
static void org_conv(const Tensor& x, int nconv, int out[3]) {
    const int p = x.pad ? nconv : 0;
    out[0] = x.org[0]; out[1] = x.org[1] - p; out[2] = x.org[2] - p;
}
static void set_org(Tensor& t, int z, int y, int x) { t.org[0] = z; t.org[1] = y; t.org[2] = x; }

// Record the branch bits of the activation tensor a launch of layer L has just written: `out` points at the launch's
// output voxel (0, 0, 0), `ext` voxels from there, whose index in the oracle's frame is `org`; periodic in y / x with the
// extent as period when `periodic`.
// zr = {lo, hi, period}: the layer's tensor exists for the planes [lo, hi) of a box that is periodic along z (the level-0
// encoder of a tile that is the whole box computes N + a few planes, down_l0 exactly N / 2; what lies outside is a periodic image)
static void probe_act(nbe_ctx* c, const Layer& L, const Planes& out, int g0, const int org[3], int ez, int ey, int ex, bool periodic,
                      const int* zr = nullptr) {
    auto& P = c->probe;
    if (!P.on || !P.tile || c->dry) return;
    const std::string name = L.block + "/" + L.layer;
    for (size_t i = 0; i < P.slots.size(); ++i) {
        const auto& S = P.slots[i];
        if (S.name != name) continue;
        ProbeLaunch a;
        a.x = out.x; a.pstride = out.pstride; a.H = out.H; a.W = out.W; a.g0 = g0; a.prec = c->prec; a.C = S.C;
        a.ext[0] = ez; a.ext[1] = ey; a.ext[2] = ex;
        a.org[0] = org[0]; a.org[1] = org[1]; a.org[2] = org[2];
        a.per[0] = 0; a.per[1] = periodic ? ey : 0; a.per[2] = periodic ? ex : 0;
        a.zlo = zr ? zr[0] : 0; a.zhi = zr ? zr[1] : 0; a.zper = zr ? zr[2] : 0;
        for (int d = 0; d < 3; ++d) a.o[d] = P.o[d] >> S.level;
        a.n = S.n; a.nw = S.nw;
        if (a.zper <= 0 && (a.o[0] + S.n <= org[0] || a.o[0] >= org[0] + ez)) return;      // this launch's planes lie outside the cone
        a.bits = P.bits + S.off; a.count = P.count + i;
        launch_probe_signs(a, c->stream);
        return;
    }
}

static int prof_entry(nbe_ctx* c, const std::string& name) {
    for (size_t i = 0; i < c->prof_entries.size(); ++i) if (c->prof_entries[i].name == name) return (int)i;
    c->prof_entries.push_back({name, 0, 0, 0});
    return (int)c->prof_entries.size() - 1;
}
static hipEvent_t get_event(nbe_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
static void prof_collect(nbe_ctx* c) {
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, p.a, p.b);
        c->prof_entries[p.entry].ms += ms;
        c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
}

static std::string conv_name(const PackedW& pw, bool vel, bool has_dx, bool g6 = false, bool up8 = false) {
    if (up8) return vel ? "up_h3<8 parities,vel,dx>" : "up_h3<8 parities,novel>";
    const char* m = pw.mode == MODE_FLAT3 ? "FLAT3" : pw.mode == MODE_FLAT1 ? "FLAT1" : "DOWN";
    const bool stem_on = !(getenv("NBE_STEM") && atoi(getenv("NBE_STEM")) == 0);
    if (stem_on && pw.stem && !(vel && has_dx) && pw.mode == MODE_FLAT3) return vel ? "stem_h3<FLAT3,vel,nodx>" : "stem_h3<FLAT3,novel>";
    char b[96];
    if (g6 && prec_is_half(pw.prec)) snprintf(b, sizeof b, "%s<%s,vel,dx>", pw.prec == PREC_F16 ? "conv_h1g" : (pw.cout_t == 16 ? "conv_h3n" : "conv_h3g"), m);
    else if (g6) snprintf(b, sizeof b, "conv_mfma_g<%s,vel,dx,ni%d>", m, pw.ni);
    else if (prec_is_half(pw.prec))
        snprintf(b, sizeof b, "%s<%s,%s,%s>", pw.prec == PREC_F16 ? "conv_h1" : "conv_h3", m, vel ? "vel" : "novel", (vel && has_dx) ? "dx" : "nodx");
    else
        snprintf(b, sizeof b, "conv_mfma<%s,%s,%s,ni%d>", m, vel ? "vel" : "novel", (vel && has_dx) ? "dx" : "nodx", pw.ni);
    return b;
}

// the float16 model's Winograd-z form (conv_h3w_kernel<., ., F16>): 32-channel stages
static bool wino_f16_layer(int prec, bool vel, int cin_pad) { return prec == PREC_F16 && vel && cin_pad % 32 == 0 && cin_pad / 32 <= 8; }
static bool wino_env_off() { return getenv("NBE_WINO") && atoi(getenv("NBE_WINO")) == 0; }   // A/B switch, read per launch
static bool narrow_off() { return getenv("NBE_NARROW") && atoi(getenv("NBE_NARROW")) == 0; }   // A/B switch (set before the context is created)
static bool narrow_tile(const Layer* L) { return L->pwn.w && !narrow_off(); }

// launch one convolution layer (or record it in a dry run)
static int run_conv(nbe_ctx* c, const Layer& L, const ConvLaunch& cl_in, bool has_dx) {
    if (c->dry) return 0;
    ConvLaunch cl = cl_in;
    // timing experiments of NBE_DBG builds (bits >= 8); production kernels ignore them
    static const int dbgf = getenv("NBE_DEBUG_FLAGS") ? atoi(getenv("NBE_DEBUG_FLAGS")) & 0xF00 : 0;
    cl.flags |= dbgf;
    const bool g6 = c->gauge_active && L.g6 && has_dx;
    const bool nov = !c->vel && c->prec == PREC_F16X3 && L.kind == 0 && L.pw.ww && c->wino_ok;   // displacement only: conv_h3w_kernel<., NOVEL>
    if (c->gauge_active) { cl.gout = L.gout; cl.beta = g6 ? L.beta : nullptr; }
    if (cl.skw) {                                                // the block's skip runs inside this launch
        if (!((g6 || nov) && c->fuse && L.fskip)) return fail("internal error: fused skip requested for %s/%s", L.block.c_str(), L.layer.c_str());
        cl.bias = L.bias_f;
    }
    const PackedW& pw = (g6 && L.kind == 0 && narrow_tile(&L)) ? L.pwn : L.pw;
    // Winograd along z (conv_h3w_kernel): gauged wide 3x3x3 launches without a fused skip or residual, on an even number
    // of output planes (the conditions of launch_h3w).  NBE_WINO=0 is the A/B switch (read per launch: tests flip it).
    // (the float16 model's form adds the residual in its epilogue: its blocks run their skips as launches of their own)
    cl.wino = (g6 || nov) && c->wino_ok && &pw == &L.pw && pw.ww && (!cl.skw || cl.skw->ww) && (c->prec == PREC_F16 || !(cl.flags & F_RES)) &&
              (cl.Dv & 1) == 0 && cl.in_off == 0 && cl.osz == 1 && !wino_env_off();
    int pe = -1; hipEvent_t ea = nullptr, eb = nullptr;
    if (c->prof) {
        std::string pn = cl.wino ? std::string(c->prec == PREC_F16 ? "conv_h1w<FLAT3,vel,dx>" : c->vel ? "conv_h3w<FLAT3,vel,dx>" : "conv_h3w<FLAT3,novel>")
                                 : conv_name(pw, c->vel, has_dx, g6, cl.set < 0);
        static const bool per_layer = getenv("NBE_PROF_LAYERS") && atoi(getenv("NBE_PROF_LAYERS")) == 1;   // tools: one entry per layer
        if (per_layer) pn += " " + L.block + "/" + L.layer;
        pe = prof_entry(c, pn);
        ea = get_event(c); eb = get_event(c);
        (void)hipEventRecord(ea, c->stream);
    }
    if (launch_conv(pw, cl, c->vel, has_dx, c->stream))
        return fail("internal error: no kernel for layer %s/%s (mode %d, gauged %d, input tangent %d, crop offset %ld, output stride %d)",
                    L.block.c_str(), L.layer.c_str(), L.pw.mode, (int)g6, (int)has_dx, (long)cl.in_off, cl.osz);
    if (c->prof) {
        (void)hipEventRecord(eb, c->stream);
        c->pending.push_back({pe, ea, eb});
        // algorithmic FLOPs: 2*MAC over valid outputs; x3 with tangent (x2 when the input has no tangent, and for
        // the gauged form W.x, W.dx~)
        const double nout = (double)cl.Dv * cl.Hv * cl.Wv * (cl.set < 0 ? 8.0 : 1.0);
        const int taps = L.kind == 0 ? 27 : L.kind == 2 ? 8 : 1;
        const double gemms = c->vel ? ((has_dx && !g6) ? 3.0 : 2.0) : 1.0;
        c->prof_entries[pe].flops += 2.0 * nout * L.cout * L.cin * taps * gemms;
        if (cl.skw) c->prof_entries[pe].flops += 2.0 * nout * L.cout * L.fskip->cin * (!c->vel ? 1.0 : (cl.flags & F_SKIP_NODX) ? 2.0 : 3.0);   // W_s.x, [W_s.dx,] dW_s.x
        c->prof_entries[pe].launches += 1;
        if (c->pending.size() > 4096) prof_collect(c);
    }
    return 0;
}

static const Layer* find_layer(nbe_ctx* c, const char* block, const char* layer) {
    auto it = c->layers.find(std::string(block) + "/" + layer);
    return it == c->layers.end() ? nullptr : &it->second;
}

// ------------------------------------------------------------------------------------------------
// schedule
// ------------------------------------------------------------------------------------------------

// StyleResNetBlock3DVel (style_blocks_vel.py:96-166): skip 1x1x1 cropped by 2, conv-act-conv, add, [act].
// Periodic-yx mode (x.pad = 1): y and x do not shrink -- every 3x3x3 convolution reads its input's wrap-around halo
// and writes the interior of a tensor of the same padded size, whose halo is filled afterwards; z shrinks as always.
// (dst: write the block's result there -- a view with the result's geometry -- instead of allocating it)
// (has_dx false: conv_l00, whose skip reads the input field -- fused with F_SKIP_NODX)
// (displacement only: conv_h3w_kernel<SKIP, NOVEL> is the one kernel that runs a fused skip without a tangent)
// (the float16 model: as displacement only -- the Winograd-z kernel is the one kernel with a fused skip)
static bool wino_only_fuse(const nbe_ctx* c) { return !c->vel || c->prec == PREC_F16; }
static bool block_fused(nbe_ctx* c, const Layer* L1, bool) { return c->fuse && L1->fskip != nullptr && (!wino_only_fuse(c) || !wino_env_off()); }

// hidden tensor of a block whose input x has `pad`: interior (Hi - sy) x (Wi - sy).  A fused block gives it the row
// and plane pitch of x (conv_h3g_kernel fetches the skip's patches of x with the offsets of its own input's).
static Tensor alloc_hidden(nbe_ctx* c, int cmid, int nz, const Tensor& x, bool fused) {
    const int pad = x.pad, sy = pad ? 0 : 2;
    if (fused && !pad) return talloc(c, cmid, nz, x.p.H, x.p.W);
    return tallocp(c, cmid, nz, x.p.H - 2 * pad - sy, x.p.W - 2 * pad - sy, pad);
}

static int resblock(nbe_ctx* c, const char* name, const Tensor& x, bool has_dx, bool final_act,
                    int cout, int cmid, Tensor* out, const Tensor* dst = nullptr) {
    const Layer *Ls = find_layer(c, name, "skip"), *L0 = find_layer(c, name, "conv_0"), *L1 = find_layer(c, name, "conv_1");
    if (!Ls || !L0 || !L1) return fail("missing layers of block %s", name);
    const int D = x.p.D, H = x.p.H, W = x.p.W, pad = x.pad;
    const int Hi = H - 2 * pad, Wi = W - 2 * pad;                 // interior of x (pad = 0: all of it)
    const int sy = pad ? 0 : 2;                                  // what one 3x3x3 convolution takes off y and x
    // (displacement only: the fused skip exists in conv_h3w_kernel alone, which pairs planes -- an odd number of result planes,
    // the 5 planes of conv_c behind a 104-voxel input, takes the unfused path)
    const bool fused = block_fused(c, L1, has_dx) && (!wino_only_fuse(c) || (D & 1) == 0);
    // Unfused: the second convolution adds the skip as a residual and writes its result over it (every lane reads its
    // residual elements before it stores the same elements): one full-resolution tensor pair less at the workspace peak.
    // Fused (gauged f16x3): conv_1 computes the skip itself from x -- no skip launch, no residual round trip.
    Tensor s = dst ? *dst : tallocp(c, cout, D - 4, Hi - 2 * sy, Wi - 2 * sy, pad);
    Tensor h = alloc_hidden(c, cmid, D - 2, x, fused);
    if ((!dst && s.off < 0) || h.off < 0) return fail("workspace exhausted in block %s", name);
    if (dst && (s.p.D != D - 4 || s.p.H != Hi - 2 * sy + 2 * pad || s.p.W != Wi - 2 * sy + 2 * pad || s.pad != pad))
        return fail("internal: destination geometry mismatch in block %s", name);
    const int64_t sk_off = (2L * H + (pad ? pad : 2)) * W + (pad ? pad : 2);   // skip: centre crop of x by the two convolutions
    if (!fused) {
        ConvLaunch cl; cl.in = x.p; cl.in_off = sk_off;
        cl.Dv = D - 4; cl.Hv = Hi - 2 * sy; cl.Wv = Wi - 2 * sy; cl.out = inner(s); cl.flags = 0;
        if (run_conv(c, *Ls, cl, has_dx)) return 1;
    }
    int og[3];
    {
        ConvLaunch cl; cl.in = x.p; cl.Dv = D - 2; cl.Hv = H - 2; cl.Wv = W - 2; cl.out = inner(h); cl.flags = F_ACT;
        if (run_conv(c, *L0, cl, has_dx)) return 1;
        org_conv(x, 1, og);
        probe_act(c, *L0, cl.out, 0, og, cl.Dv, cl.Hv, cl.Wv, pad != 0);
    }
    fill_halo(c, h);
    org_conv(x, 2, og);
    {
        ConvLaunch cl; cl.in = h.p; cl.Dv = D - 4; cl.Hv = s.p.H - 2 * pad; cl.Wv = s.p.W - 2 * pad; cl.out = inner(s);
        if (fused) { cl.sk = x.p; cl.sk_off = sk_off; cl.skw = narrow_tile(L1) ? &Ls->pwn : &Ls->pw; cl.flags = (final_act ? F_ACT : 0) | (has_dx ? 0 : F_SKIP_NODX); }
        else { cl.res = inner(s); cl.flags = F_RES | (final_act ? F_ACT : 0); }
        if (run_conv(c, *L1, cl, true)) return 1;
        if (final_act) probe_act(c, *L1, cl.out, 0, og, cl.Dv, cl.Hv, cl.Wv, pad != 0);
    }
    fill_halo(c, s);
    tfree(c, h);
    set_org(s, og[0], og[1], og[2]);
    *out = s;
    return 0;
}

// The z-slab schedule's residual block: the same three launches as resblock() on plane ranges of persistent tensors.
// Plane indices are in block-input coordinates (result plane j is centred on input plane j + 2): hidden planes
// [jh, jh + nh) and result planes [js, js + ns) are computed; what precedes them was carried over from the slab
// before.  h and s have the geometry resblock() would give them (s also serves as the skip / residual, in place).
// x2: the block input is concat([x, x2]) along the channels (mid channels each, same geometry) without a concat tensor:
// the gauged f16x3 kernel reads its K chunks from two tensors (fused blocks only)
static int resblock_part(nbe_ctx* c, const char* name, const Tensor& x, const Tensor& h, const Tensor& s,
                         int js, int ns, int jh, int nh, bool has_dx, bool final_act, const Tensor* x2 = nullptr) {
    const Layer *Ls = find_layer(c, name, "skip"), *L0 = find_layer(c, name, "conv_0"), *L1 = find_layer(c, name, "conv_1");
    if (!Ls || !L0 || !L1) return fail("missing layers of block %s", name);
    const int H = x.p.H, W = x.p.W, pad = x.pad;
    const bool fused = block_fused(c, L1, has_dx);
    if (fused && (h.p.H != H || h.p.W != W)) return fail("internal: hidden tensor of fused block %s lacks the input's pitch", name);
    if (x2 && (!fused || x2->p.H != H || x2->p.W != W || x2->pad != pad)) return fail("internal: two-source input of block %s", name);
    const Tensor sv = zview(s, js, ns), hv = zview(h, jh, nh);
    const Tensor xs = zview(x, js, ns + 4);                      // what the skip of result planes [js, js + ns) reads
    const int64_t sk_off = (2L * H + (pad ? pad : 2)) * W + (pad ? pad : 2);
    if (!fused) {
        ConvLaunch cl; cl.in = xs.p; cl.in_off = sk_off;
        cl.Dv = ns; cl.Hv = s.p.H - 2 * pad; cl.Wv = s.p.W - 2 * pad; cl.out = inner(sv); cl.flags = 0;
        if (run_conv(c, *Ls, cl, has_dx)) return 1;
    }
    int og[3];
    {
        ConvLaunch cl; cl.in = zview(x, jh, nh + 2).p; cl.Dv = nh; cl.Hv = H - 2; cl.Wv = W - 2; cl.out = inner(hv); cl.flags = F_ACT;
        if (x2) { cl.in2 = zview(*x2, jh, nh + 2).p; cl.csplit_ch = c->mid; }
        if (run_conv(c, *L0, cl, has_dx)) return 1;
        org_conv(x, 1, og); og[0] += jh;                         // hidden plane j is centred on plane j + 1 of x
        const int zh[3] = {c->probe.zr[0], c->probe.zr[1] + 2, c->probe.zr[2]};
        probe_act(c, *L0, cl.out, 0, og, cl.Dv, cl.Hv, cl.Wv, pad != 0, zh);
    }
    fill_halo(c, hv);
    {
        ConvLaunch cl; cl.in = zview(h, js, ns + 2).p; cl.Dv = ns; cl.Hv = s.p.H - 2 * pad; cl.Wv = s.p.W - 2 * pad; cl.out = inner(sv);
        if (fused) { cl.sk = xs.p; cl.sk_off = sk_off; cl.skw = narrow_tile(L1) ? &Ls->pwn : &Ls->pw; cl.flags = (final_act ? F_ACT : 0) | (has_dx ? 0 : F_SKIP_NODX);
                     if (x2) { cl.sk2 = zview(*x2, js, ns + 4).p; cl.sk_split_ch = c->mid; } }
        else { cl.res = inner(sv); cl.flags = F_RES | (final_act ? F_ACT : 0); }
        if (run_conv(c, *L1, cl, true)) return 1;
        org_conv(x, 2, og); og[0] += js;
        if (final_act) probe_act(c, *L1, cl.out, 0, og, cl.Dv, cl.Hv, cl.Wv, pad != 0, c->probe.zr);
    }
    fill_halo(c, sv);
    return 0;
}

// planes [src, src + n) of t -> planes [dst, dst + n) of the same tensor (the ranges must not overlap)
static void carry_planes(nbe_ctx* c, const Tensor& t, int src, int dst, int n) {
    if (c->dry || n <= 0) return;
    launch_crop(zview(t, src, n).p, 0, zview(t, dst, n).p, 0, c->vel, c->stream, 0);
}

static int downblock(nbe_ctx* c, const char* name, const Tensor& x, Tensor* out) {
    const Layer* L = find_layer(c, name, "conv_0");
    if (!L) return fail("missing layer %s/conv_0", name);
    Tensor o = talloc(c, c->mid, x.p.D / 2, x.p.H / 2, x.p.W / 2);
    if (o.off < 0) return fail("workspace exhausted in %s", name);
    ConvLaunch cl; cl.in = x.p; cl.Dv = o.p.D; cl.Hv = o.p.H; cl.Wv = o.p.W; cl.out = o.p; cl.flags = F_ACT;
    if (run_conv(c, *L, cl, true)) return 1;
    set_org(o, x.org[0] / 2, x.org[1] / 2, x.org[2] / 2);
    probe_act(c, *L, cl.out, 0, o.org, cl.Dv, cl.Hv, cl.Wv, false);
    *out = o;
    return 0;
}

// up-sample into planes [mid/4, 2*mid/4) of the concat tensor (core :166-169: concat([skip, up]))
// (xcrop: centre crop of x in y and x before up-sampling; the result goes to the interior of cat)
// (g0 < 0: into the second half of a 2 * mid channel concat tensor; g0 = 0: into a mid channel tensor of its own)
static int upblock(nbe_ctx* c, const char* name, const Tensor& x, const Tensor& cat, int xcrop = 0, int g0 = -1) {
    const Layer* L = find_layer(c, name, "conv_0");
    if (!L) return fail("missing layer %s/conv_0", name);
    xcrop += x.pad;                                              // a periodic halo of x is not up-sampled either
    const int Hx = x.p.H - 2 * xcrop, Wx = x.p.W - 2 * xcrop;
    if (cat.p.D != 2 * x.p.D || cat.p.H - 2 * cat.pad != 2 * Hx || cat.p.W - 2 * cat.pad != 2 * Wx)
        return fail("internal: concat geometry mismatch in %s", name);
    // f16-based arithmetic and Cin <= 64: all eight parities in one launch (up_h3_kernel: the input is read once)
    const bool up8_off = getenv("NBE_UP8") && atoi(getenv("NBE_UP8")) == 0;                 // A/B switch
    const bool up8 = prec_is_half(c->prec) && L->pw.cin_pad <= 64 && !up8_off;
    const int out_g0 = g0 >= 0 ? g0 : c->mid / (c->prec == PREC_F16 ? 8 : 4);
    for (int p = 0; p < (up8 ? 1 : 8); ++p) {
        ConvLaunch cl; cl.in = x.p; cl.in_off = ((int64_t)xcrop * x.p.W + xcrop);
        cl.Dv = x.p.D; cl.Hv = Hx; cl.Wv = Wx; cl.out = inner(cat);
        cl.out_g0 = out_g0; cl.osz = 2; cl.oz = (p >> 2) & 1; cl.oy = (p >> 1) & 1; cl.ox = p & 1;
        cl.flags = F_ACT; cl.set = up8 ? -1 : p;
        if (run_conv(c, *L, cl, true)) return 1;
    }
    // output voxel 2 i + parity comes from the input voxel i of the cropped interior (the crop beyond x's own halo)
    const int xc = xcrop - x.pad;
    const int og[3] = {2 * x.org[0], 2 * (x.org[1] + xc), 2 * (x.org[2] + xc)};
    probe_act(c, *L, inner(cat), out_g0, og, 2 * x.p.D, 2 * Hx, 2 * Wx, cat.pad != 0);
    return 0;
}

static void crop_into(nbe_ctx* c, const Tensor& src, int crop, const Tensor& cat) {
    if (c->dry) return;
    Planes s = src.p; s.G = c->mid / (c->prec == PREC_F16 ? 8 : 4);
    launch_crop(s, crop, cat.p, 0, c->vel, c->stream);
}

static int check_dims(int D, int H, int W) {
    const int v[3] = {D, H, W};
    for (int i = 0; i < 3; ++i)
        if (v[i] < 104 || v[i] % 8 != 0)
            return fail("input spatial size %d unsupported: each of (D,H,W) must be >= 104 and a multiple of 8 "
                        "(all-VALID U-Net with three 2x levels, receptive-field crop 48)", v[i]);
    return 0;
}

// the network body on a resident input tensor; returns conv_r01's output tensor (out_chan channels)
static int network(nbe_ctx* c, const Tensor& tin, Tensor* yout) {
    const int m = c->mid;
    Tensor a, y0, y1, y2, t, cat0, cat1, cat2, r;
    if (resblock(c, "conv_l00", tin, false, true, m, m, &a)) return 1;
    if (resblock(c, "conv_l01", a, true, true, m, m, &y0)) return 1;
    tfree(c, a);
    cat0 = talloc(c, 2 * m, y0.p.D - 80, y0.p.H - 80, y0.p.W - 80);
    if (cat0.off < 0) return fail("workspace exhausted (cat0)");
    crop_into(c, y0, 40, cat0);
    if (downblock(c, "down_l0", y0, &t)) return 1;
    tfree(c, y0);

    if (resblock(c, "conv_l1", t, true, true, m, m, &y1)) return 1;
    tfree(c, t);
    cat1 = talloc(c, 2 * m, y1.p.D - 32, y1.p.H - 32, y1.p.W - 32);
    if (cat1.off < 0) return fail("workspace exhausted (cat1)");
    crop_into(c, y1, 16, cat1);
    if (downblock(c, "down_l1", y1, &t)) return 1;
    tfree(c, y1);

    if (resblock(c, "conv_l2", t, true, true, m, m, &y2)) return 1;
    tfree(c, t);
    cat2 = talloc(c, 2 * m, y2.p.D - 8, y2.p.H - 8, y2.p.W - 8);
    if (cat2.off < 0) return fail("workspace exhausted (cat2)");
    crop_into(c, y2, 4, cat2);
    if (downblock(c, "down_l2", y2, &t)) return 1;
    tfree(c, y2);

    if (resblock(c, "conv_c", t, true, true, m, m, &r)) return 1;
    tfree(c, t);

    if (upblock(c, "up_r2", r, cat2)) return 1;
    tfree(c, r);
    if (resblock(c, "conv_r2", cat2, true, true, m, 2 * m, &r)) return 1;
    tfree(c, cat2);

    if (upblock(c, "up_r1", r, cat1)) return 1;
    tfree(c, r);
    if (resblock(c, "conv_r1", cat1, true, true, m, 2 * m, &r)) return 1;
    tfree(c, cat1);

    if (upblock(c, "up_r0", r, cat0)) return 1;
    tfree(c, r);
    if (resblock(c, "conv_r00", cat0, true, true, m, 2 * m, &r)) return 1;
    tfree(c, cat0);

    if (resblock(c, "conv_r01", r, true, false, c->out_chan, m, yout)) return 1;
    tfree(c, r);
    return 0;
}

// Where the head writes: the (C, OD, OH, OW) output boxes and the anchor of this tile in them.
struct HeadOut { void* disp; void* velo; int out_dtype; int OD, OH, OW, a0, a1, a2; float Dz, vel_fac; };

static void run_head(nbe_ctx* c, const Tensor& y, const Tensor& xin, const HeadOut& h, int zoff) {
    if (c->dry) return;
    // core :187-193 with the call's range shift s = 2^k divided out (exact): disp = (y + x0) * 6 / s,
    // vel = dy * (vf * 6 / s) + x0 * (vf * 6 / (Dz * s))
    const float inv_s = 1.0f / c->act_scale;
    HeadScale hs;
    hs.k_disp = 6.0f * inv_s; hs.k_dy = h.vel_fac * 6.0f * inv_s; hs.k_x0 = h.vel_fac * 6.0f / h.Dz * inv_s;
    hs.bad = c->flags ? (int*)(c->flags + 1) : nullptr;
    launch_head(y.p, xin.p, 48, c->out_chan, hs, c->vel, h.disp, h.velo, h.out_dtype, h.OD, h.OH, h.OW,
                h.a0 + zoff, h.a1, h.a2, c->prec, c->stream, y.pad);
}

// ------------------------------------------------------------------------------------------------
// Range shift (include/nbe.h, "Range").  LeakyReLU is positively homogeneous and the convolutions are linear, so the
// network with input x and biases b satisfies f(s x; s b) = s f(x; b) for s > 0, exactly in floating point when s is a
// power of two.  The f16-based modes use that to keep their operands where f16 has both range and precision: s = 2^k
// brings max(|x| Dz / 6, max |b|) into [0.5, 1) (float16) or [2^5, 2^6) (f16x3: H3_RANGE_UP) whatever the caller's units are.
// ------------------------------------------------------------------------------------------------
static int prepare_range(nbe_ctx* c, const float* dev_src, int64_t n, float Dz, const float* host_src = nullptr) {
    c->input_finite = true;
    float s = 1.f;
    if (prec_is_half(c->prec)) {
        if (!c->flags) { HIPCHK(hipMalloc((void**)&c->flags, 8)); HIPCHK(hipMemsetAsync(c->flags, 0, 8, c->stream)); }
        float amax = c->preset_absmax;
        if (amax < 0.f && host_src) {                           // pipelined host path: the box is not on the device yet
            const unsigned bits = host_absmax_bits(host_src, n);
            memcpy(&amax, &bits, 4);
        } else if (amax < 0.f) {
            HIPCHK(hipMemsetAsync(c->flags, 0, 4, c->stream));
            launch_absmax(dev_src, n, c->flags, c->stream);
            unsigned bits = 0;
            HIPCHK(hipMemcpyAsync(&bits, c->flags, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            memcpy(&amax, &bits, 4);
        }
        if (!std::isfinite(amax)) c->input_finite = false;     // NaN / infinity in: NaN / infinity out, as in the reference
        else {
            const float m = std::max(amax * std::fabs(Dz) / 6.0f, c->bias_max);
            if (m > 0.f && std::isfinite(m)) {
                int e = 0;
                (void)std::frexp(m, &e);                        // m = f * 2^e, f in [0.5, 1)
                e = std::max(-100, std::min(100, e));
                s = std::ldexp(1.0f, -e + (c->prec == PREC_F16X3 ? H3_RANGE_UP : 0));
            }
        }
    }
    c->act_scale = s;
    if (s != c->bias_scale || c->bias_dirty) {
        for (auto& kv : c->layers) {
            Layer& L = kv.second;
            const int nb = L.pw.ctiles * 32 * L.pw.ni;
            launch_scale(L.bias0, L.pw.bias, nb, s, c->stream);
            if (L.fskip) launch_scale(L.bias0, L.bias_f, nb, s, c->stream, L.fskip->bias0);   // fused block: b_1 + b_s
        }
        c->bias_scale = s; c->bias_dirty = false;
    }
    c->range_pending = true;
    return 0;
}

// after a call: did the head write a non-finite value although the input was finite?  (synchronises the stream)
static int check_range(nbe_ctx* c) {
    if (!c->range_pending || !c->flags) { c->range_pending = false; return 0; }
    c->range_pending = false;
    unsigned bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, c->flags + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemsetAsync(c->flags + 1, 0, 4, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (bad & 2u)
        return fail("brick mode: a neighbour's faces were computed with another range shift -- every rank must call "
                    "nbe_set_input_range with the box-wide max |x| before nbe_brick_encode");
    if (bad && c->input_finite) {
        fail("non-finite values in the output of a finite input: an activation left the range of the %s arithmetic "
             "(|value| >= 65504 * 2^%d after the range shift); rerun this call with NBE_PREC_F32",
             c->prec == PREC_F16 ? "float16" : "f16x3", -(int)std::lround(std::log2(c->act_scale)));
        return 2;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Pipelined host path (HostPipe).  The tile is the whole periodic box; tile plane t is box plane (o0 + t) mod S0.
// ------------------------------------------------------------------------------------------------
static constexpr int PIPE_CHUNK = 32;                            // box planes per staged upload
static constexpr int PIPE_EDGE = 32;                             // planes of the first encoder slab and of the last decoder slab

// box planes behind tile planes [t0, t1) -> device box (enqueued on up_stream; pageable sources go through the pinned
// staging ring, filled by host threads)
static int pipe_upload(nbe_ctx* c, int t0, int t1) {
    auto& P = c->pipe;
    const int64_t plane = (int64_t)P.S1 * P.S2;
    const int D = P.S0 + 96;                                      // (a tile is at most the box + its halo deep)
    t0 = std::max(t0, 0); t1 = std::min(t1, D);
    int t = t0;
    while (t < t1) {
        const int b = ((P.o0 + t) % P.S0 + P.S0) % P.S0;
        if (P.up[b]) { ++t; continue; }
        int run = 1;
        while (t + run < t1 && b + run < P.S0 && !P.up[b + run] && run < PIPE_CHUNK) ++run;
        const size_t bytes = (size_t)run * plane * 4;            // per channel
        float* dbox = c->box_in;
        if (P.in_pinned) {
            for (int ch = 0; ch < P.C; ++ch)
                HIPCHK(hipMemcpyAsync(dbox + ((int64_t)ch * P.S0 + b) * plane, P.hbox + ((int64_t)ch * P.S0 + b) * plane,
                                      bytes, hipMemcpyHostToDevice, c->up_stream));
        } else {
            const int slot = P.nstage % nbe_ctx::NSTAGE;
            if (P.nstage >= nbe_ctx::NSTAGE) HIPCHK(hipEventSynchronize(c->stage_free[slot]));   // its last DMA has finished
            for (int ch = 0; ch < P.C; ++ch)
                parallel_memcpy(c->stage_buf[slot] + ch * bytes, P.hbox + ((int64_t)ch * P.S0 + b) * plane, bytes);
            for (int ch = 0; ch < P.C; ++ch)
                HIPCHK(hipMemcpyAsync(dbox + ((int64_t)ch * P.S0 + b) * plane, c->stage_buf[slot] + ch * bytes, bytes,
                                      hipMemcpyHostToDevice, c->up_stream));
            HIPCHK(hipEventRecord(c->stage_free[slot], c->up_stream));
            ++P.nstage;
        }
        for (int k = 0; k < run; ++k) P.up[b + k] = 1;
        t += run;
    }
    return 0;
}

// tile planes [t0, t1) of the input tensor: upload what is missing, gather them (core :132-134 scaling), then start the
// upload of the `look` planes that follow so that it runs under the kernels enqueued next
static int pipe_input(nbe_ctx* c, const Tensor& tin, int t0, int t1, int look, float scale) {
    auto& P = c->pipe;
    if (P.gz < t0) P.gz = t0;                                    // planes before t0 are never read
    if (t1 > P.gz) {
        if (pipe_upload(c, P.gz, t1)) return 1;
        HIPCHK(hipEventRecord(c->ev_up, c->up_stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_up, 0));
        const Tensor v = zview(tin, P.gz, t1 - P.gz);
        const int h = tin.pad ? 1 : 48;
        launch_gather(c->box_in, P.C, P.S0, P.S1, P.S2, P.o0 + P.gz, P.slabwise ? P.o1 : -h, P.slabwise ? P.o2 : -h, v.p, scale,
                      c->prec, c->stream);
        P.gz = t1;
    }
    return look > 0 ? pipe_upload(c, t1, t1 + look) : 0;
}

// output planes [z, z + n) of every channel of both fields: device staging -> the caller's pinned arrays, on the
// down stream, behind the head launch that produced them
static int pipe_output(nbe_ctx* c, int z, int n, int a1 = 0, int a2 = 0, int e1 = -1, int e2 = -1) {
    auto& P = c->pipe;
    if (e1 < 0) e1 = P.O1;
    if (e2 < 0) e2 = P.O2;
    HIPCHK(hipEventRecord(c->ev_down, c->stream));
    HIPCHK(hipStreamWaitEvent(c->down_stream, c->ev_down, 0));
    const int64_t plane = (int64_t)P.O1 * P.O2 * P.esz;
    const bool whole = a1 == 0 && a2 == 0 && e1 == P.O1 && e2 == P.O2;
    for (int f = 0; f < (P.hvel ? 2 : 1); ++f) {
        char* h = f ? P.hvel : P.hdisp;
        char* d = f ? P.dvel : P.ddisp;
        for (int ch = 0; ch < c->out_chan; ++ch) {
            if (whole) {
                const int64_t off = ((int64_t)ch * P.O0 + z) * plane;
                HIPCHK(hipMemcpyAsync(h + off, d + off, (size_t)n * plane, hipMemcpyDeviceToHost, c->down_stream));
            } else {                                             // a tile's (n, e1, e2) block of the (C, O0, O1, O2) arrays
                hipMemcpy3DParms mp;
                memset(&mp, 0, sizeof mp);
                mp.srcPtr = make_hipPitchedPtr(d, (size_t)P.O2 * P.esz, (size_t)P.O2 * P.esz, (size_t)P.O1);
                mp.dstPtr = make_hipPitchedPtr(h, (size_t)P.O2 * P.esz, (size_t)P.O2 * P.esz, (size_t)P.O1);
                mp.srcPos = make_hipPos((size_t)a2 * P.esz, (size_t)a1, (size_t)ch * P.O0 + z);
                mp.dstPos = mp.srcPos;
                mp.extent = make_hipExtent((size_t)e2 * P.esz, (size_t)e1, (size_t)n);
                mp.kind = hipMemcpyDeviceToHost;
                HIPCHK(hipMemcpy3DAsync(&mp, c->down_stream));
            }
        }
    }
    return 0;
}

// The same network with the two full-resolution levels run in slabs of S output planes (S even): the encoder blocks
// conv_l00 / conv_l01 (+ the crop of the skip connection and down_l0) and the decoder blocks up_r0 / conv_r00 /
// conv_r01 (+ head) only ever hold slab-sized tensors, so a tile can be as deep as the box (no halo recompute along
// z inside it) at a fraction of the workspace.  Neighbouring slabs recompute the 2-plane overlaps of the 3x3x3
// layers: (S + 6) / S on the first hidden tensor, less further down.  Everything is the whole-tensor schedule on
// z-views of the same tensors; results are identical.
//
// Periodic-yx mode (tin.pad = 1: the tile spans the whole periodic box in y and x).  The two full-resolution levels
// do not pad-and-shrink in y and x: their tensors are N + 2 wide, every 3x3x3 convolution reads the wrap-around halo
// of its input and the halo of its output is filled afterwards -- the same arithmetic per voxel as the reference's
// 48-voxel periodic padding, without computing the halo voxels (about 10 % of the FLOPs of a 512^3 box).  The levels
// below keep the padded scheme: down_l0 runs on the interior and its output is extended periodically by the 22
// voxels of context those levels consume; up_r0 takes the centre of the level-1 result.
static int stream_encode(nbe_ctx* c, const Tensor& tin, const HeadOut& ho, int S, Tensor* skip0_out, Tensor* td_out) {
    const int m = c->mid, pad = tin.pad;
    const int D = tin.p.D, H = tin.p.H, W = tin.p.W;
    const int Hi = H - 2 * pad, Wi = W - 2 * pad;
    const int Y = D - 8;                                          // planes of the level-0 encoder output
    // Periodic in z too (the tile is the whole box): the 40 outermost planes of the level-0 encoder output on either
    // side only feed the lower levels, whose input can be extended periodically in z just as in y and x.  The encoder
    // then produces the Y - 80 planes of the skip connection only, and down_l0 the box's own (D - 96) / 2 planes.
    const bool zx = pad && c->zx;                                 // brick mode: as pz, the z context of level 1 comes from the neighbours
    const bool pz = pad && (c->pz || zx);
    // brick mode: the brick's own D - 96 planes of the skip connection only (planes 4 .. of the tensor) -- the four on either
    // side that the decoder reads as well are the neighbours' own planes and arrive by exchange (network_stream)
    const int zlo = pz ? (zx ? 44 : 40) : 0, zhi = pz ? (zx ? Y - 44 : Y - 40) : Y;
    // the level-0 skip connection: centre crop by 40 (z only in periodic-yx mode)
    Tensor skip0 = pad ? tallocp(c, m, Y - 80, Hi, Wi, pad) : talloc(c, m, Y - 80, H - 88, W - 88);
    // down_l0 output; periodic-yx: on the interior first (td), then extended by 22 voxels of periodic context (t)
    Tensor td = pad ? talloc(c, m, pz ? (D - 96) / 2 : Y / 2, Hi / 2, Wi / 2) : talloc(c, m, Y / 2, (H - 8) / 2, (W - 8) / 2);
    if (skip0.off < 0 || td.off < 0) return fail("workspace exhausted (level 0)");
    const Layer* Ld = find_layer(c, "down_l0", "conv_0");
    if (!Ld) return fail("missing layer down_l0/conv_0");
    // Persistent slab tensors of the level-0 encoder: hidden and result of conv_l00 (h0, a), hidden of conv_l01 (h1)
    // and, unless the slabs land in the skip tensor directly, its result (y0).  Consecutive slabs overlap by 6 / 4 / 2
    // planes of h0 / a / h1: those are carried over from the slab before (a copy of a few planes) instead of being
    // recomputed, so every layer computes every plane exactly once.
    const int sy = pad ? 0 : 2;
    const Layer* L00 = find_layer(c, "conv_l00", "conv_1");
    if (!L00) return fail("missing layer conv_l00/conv_1");
    Tensor h0 = alloc_hidden(c, m, S + 6, tin, block_fused(c, L00, false)), a = tallocp(c, m, S + 4, Hi - 2 * sy, Wi - 2 * sy, pad);
    const Layer *L01 = find_layer(c, "conv_l01", "conv_1"), *Lr00 = find_layer(c, "conv_r00", "conv_1"), *Lr01 = find_layer(c, "conv_r01", "conv_1");
    if (!L01 || !Lr00 || !Lr01) return fail("missing conv_1 layers of the level-0 blocks");
    Tensor h1 = alloc_hidden(c, m, S + 2, a, block_fused(c, L01, true));
    Tensor y0r = pz ? Tensor() : tallocp(c, m, S, Hi - 4 * sy, Wi - 4 * sy, pad);
    if (h0.off < 0 || a.off < 0 || h1.off < 0 || (!pz && y0r.off < 0)) return fail("workspace exhausted (level-0 encoder slabs)");
    // Pipelined host path: the first slab is short (PIPE_EDGE planes), so that the kernels start as soon as a small first
    // upload has landed; the decoder's last slab is short for the same reason at the other end (its copy to the host is
    // the only one nothing hides).  Slabs start on even planes either way, so the fields do not change.
    for (int z = zlo, n = 0; z < zhi; z += n) {
        n = std::min(((c->pipe.active || c->pipe.slabwise) && z == zlo) ? std::min(S, PIPE_EDGE) : S, zhi - z);
        const int n_next = std::min(S, zhi - (z + n));
        const bool first = z == zlo;
        // periodic in z: the slab is exactly planes [z - 40, z - 40 + n) of the skip connection -- write it there
        Tensor y0 = pz ? zview(skip0, z - 40, n) : zview(y0r, 0, n);
        if ((c->pipe.active || c->pipe.slabwise) && !c->dry && pipe_input(c, tin, z, z + n + 8, n_next, ho.Dz / 6.0f * c->act_scale)) return 1;
        // frames (branch probe): plane j of the persistent slab tensors is plane z + j of the layer's whole tensor
        { int og[3]; org_conv(zview(tin, z, n + 8), 2, og); set_org(a, og[0], og[1], og[2]);
          org_conv(a, 2, og); set_org(y0, og[0], og[1], og[2]); }
        // (branch probe: periodic in z, the planes [zlo, zhi + 4) of conv_l00's result and [zlo, zhi) of conv_l01's exist)
        auto zr = [&](int extra) { c->probe.zr[0] = zlo; c->probe.zr[1] = zhi + extra; c->probe.zr[2] = pz ? D - 96 : 0; };
        if (first) {
            zr(4); if (resblock_part(c, "conv_l00", zview(tin, z, n + 8), h0, a, 0, n + 4, 0, n + 6, false, true)) return 1;
            zr(0); if (resblock_part(c, "conv_l01", a, h1, y0, 0, n, 0, n + 2, true, true)) return 1;
        } else {
            zr(4); if (resblock_part(c, "conv_l00", zview(tin, z, n + 8), h0, a, 4, n, 6, n, false, true)) return 1;
            zr(0); if (resblock_part(c, "conv_l01", a, h1, y0, 0, n, 2, n, true, true)) return 1;
        }
        c->probe.zr[2] = 0;
        if (z + n < zhi) {                                       // what the next slab will not recompute
            carry_planes(c, h0, n, 0, 6);
            carry_planes(c, a, n, 0, 4);
            carry_planes(c, h1, n, 0, 2);
        }
        const int i0 = std::max(0, 40 - z), i1 = std::min(n, Y - 40 - z);      // planes of this slab inside the crop
        if (!pz && i1 > i0 && !c->dry) {
            Planes sp = y0.p; sp.G = c->mid / (c->prec == PREC_F16 ? 8 : 4);
            launch_crop(sp, pad ? 0 : 40, zview(skip0, z + i0 - 40, i1 - i0).p, 0, c->vel, c->stream, i0);
        }
        {
            // planes [d0, d1) of this slab go through down_l0 (periodic in z: only the box's own planes, 44 .. Y - 44)
            const int d0 = pz ? std::max(z, 44) : z, d1 = pz ? std::min(z + n, Y - 44) : z + n;
            if (d1 > d0) {
                const Tensor tv = zview(td, (d0 - (pz ? 44 : 0)) / 2, (d1 - d0) / 2);
                Tensor yv = zview(y0, d0 - z, d1 - d0);
                ConvLaunch cl; cl.in = inner(yv); cl.Dv = tv.p.D; cl.Hv = tv.p.H; cl.Wv = tv.p.W; cl.out = tv.p; cl.flags = F_ACT;
                if (run_conv(c, *Ld, cl, true)) return 1;
                const int og[3] = {yv.org[0] / 2, yv.org[1] / 2, yv.org[2] / 2};
                const int zd[3] = {22, 22 + (D - 96) / 2, pz ? (D - 96) / 2 : 0};   // periodic in z: the box's own N / 2 planes
                probe_act(c, *Ld, cl.out, 0, og, cl.Dv, cl.Hv, cl.Wv, pad != 0, zd);     // periodic-yx: down_l0 ran on the interior only
            }
        }
    }
    tfree(c, h0); tfree(c, a); tfree(c, h1);
    if (!pz) tfree(c, y0r);
    // frames: the skip connection is conv_l01's result cropped by 40 (its frame starts 40 voxels in; periodic-yx keeps all
    // of y and x, whose interior sits 44 voxels into the padded frame); down_l0's output starts at plane 44 / 2 when the
    // encoder only produced the box's own planes (pz)
    set_org(skip0, 0, pad ? 4 : 0, pad ? 4 : 0);
    set_org(td, pz ? 22 : 0, pad ? 22 : 0, pad ? 22 : 0);
    *skip0_out = skip0; *td_out = td;
    return 0;
}

// Level 1 of the encoder outside brick mode: the down_l0 output td -> the level-1 skip connection cat1 (first half of the
// decoder's concat) and the level-2 input t.
// Periodic-yx: level 1 runs periodic in y and x as well -- its input is the interior result of down_l0 with a 1-voxel
// wrap-around halo (and, periodic in z, 22 planes of periodic context); level 2 and below keep the padded scheme: down_l1
// runs on the interior and is extended periodically by the 10 voxels those levels consume.
static int stream_level1(nbe_ctx* c, int pad, bool pz, Tensor td, Tensor* cat1_out, Tensor* t_out) {
    const int m = c->mid;
    Tensor t = td;
    if (pad) {
        t = tallocp(c, m, td.p.D + (pz ? 44 : 0), td.p.H, td.p.W, 1);
        if (t.off < 0) return fail("workspace exhausted (level 1 input)");
        if (!c->dry) launch_wrap_pad(td.p, t.p, 1, c->vel, c->stream, pz ? 22 : 0);
        set_org(t, pz ? td.org[0] - 22 : td.org[0], td.org[1], td.org[2]);
        tfree(c, td);
    }

    Tensor y1, cat1;
    if (resblock(c, "conv_l1", t, true, true, m, m, &y1)) return 1;
    tfree(c, t);
    if (pad) {
        cat1 = tallocp(c, 2 * m, y1.p.D - 32, y1.p.H - 2, y1.p.W - 2, 1);
        if (cat1.off < 0) return fail("workspace exhausted (cat1)");
        if (!c->dry) {
            Planes sp = y1.p; sp.G = c->mid / (c->prec == PREC_F16 ? 8 : 4);
            launch_crop(sp, 0, cat1.p, 0, c->vel, c->stream, 16);
        }
        set_org(cat1, y1.org[0], y1.org[1] - 16, y1.org[2] - 16);    // cropped by 16 in z only; the frame moves by 16 on every axis
        Tensor t2 = talloc(c, m, y1.p.D / 2, (y1.p.H - 2) / 2, (y1.p.W - 2) / 2);
        const Layer* Ld1 = find_layer(c, "down_l1", "conv_0");
        if (t2.off < 0 || !Ld1) return fail("workspace exhausted or missing layer (down_l1)");
        ConvLaunch cl; cl.in = inner(y1); cl.Dv = t2.p.D; cl.Hv = t2.p.H; cl.Wv = t2.p.W; cl.out = t2.p; cl.flags = F_ACT;
        if (run_conv(c, *Ld1, cl, true)) return 1;
        set_org(t2, y1.org[0] / 2, y1.org[1] / 2, y1.org[2] / 2);
        probe_act(c, *Ld1, cl.out, 0, t2.org, cl.Dv, cl.Hv, cl.Wv, true);   // the interior; its periodic images are copies (wrap_pad below)
        t = talloc(c, m, t2.p.D, t2.p.H + 20, t2.p.W + 20);
        if (t.off < 0) return fail("workspace exhausted (level 2 input)");
        if (!c->dry) launch_wrap_pad(t2.p, t.p, 10, c->vel, c->stream, 0);
        set_org(t, t2.org[0], t2.org[1] - 10, t2.org[2] - 10);
        tfree(c, t2);
    } else {
        cat1 = talloc(c, 2 * m, y1.p.D - 32, y1.p.H - 32, y1.p.W - 32);
        if (cat1.off < 0) return fail("workspace exhausted (cat1)");
        crop_into(c, y1, 16, cat1);
        if (downblock(c, "down_l1", y1, &t)) return 1;
    }
    tfree(c, y1);
    *cat1_out = cat1; *t_out = t;
    return 0;
}

// ---- brick mode (one rank's z-slab of a periodic box; include/nbe.h, "Brick mode") --------------------------------------
// What a brick needs from its z neighbours below the full-resolution level is exchanged instead of recomputed, at the two
// places where it is smallest: BRICK_H1 planes of the down_l0 output per side (what conv_l1 reads beyond the brick's own
// planes for the level-1 skip connection: 4 + 2) and BRICK_H2 planes of the down_l1 output (what levels 2 and 3 read: 10).
// Own planes of the level-1 input sit at [BRICK_H1, BRICK_H1 + B) of t.
static constexpr int BRICK_H1 = 6, BRICK_H2 = 10;
// ... and at the full-resolution level: BRICK_H0 planes of the skip connection (conv_l01's output) per side, which the decoder's
// first block reads beyond the brick's own planes -- exchanged while levels 1-3 run, instead of 8 more planes through the four
// layers of the level-0 encoder
static constexpr int BRICK_H0 = 4;
static Planes brick_planes(nbe_ctx* c, const Tensor& like, const void* buf, int nplanes) {
    Planes p = like.p;
    p.D = nplanes;
    p.pstride = (p.vox() + 63) & ~int64_t(63);
    p.x = (float*)buf;
    p.dx = c->vel ? (float*)buf + (int64_t)p.G * p.pstride * 4 : nullptr;
    return p;
}
static int64_t brick_halo_bytes(nbe_ctx* c, int nplanes, int Hd, int Wd) {
    Planes p; p.G = planes_for(c->mid, c->prec); p.D = nplanes; p.H = Hd; p.W = Wd;
    p.pstride = (p.vox() + 63) & ~int64_t(63);
    return (int64_t)p.G * p.pstride * 16 * (c->vel ? 2 : 1);
}

// conv_l1 on plane ranges of the whole level-1 tensors (resblock_part): part 0 = what depends on the brick's own planes only,
// part 1 = the planes next to the low face, part 2 = next to the high face
static int brick_conv_l1(nbe_ctx* c, const Tensor& t, const Tensor& h, const Tensor& y1, int part) {
    const int B = t.p.D - 2 * BRICK_H1;
    if (part == 0) return resblock_part(c, "conv_l1", t, h, y1, BRICK_H1, B - 4, BRICK_H1, B - 2, true, true);
    if (part == 1) return resblock_part(c, "conv_l1", t, h, y1, 0, BRICK_H1, 0, BRICK_H1, true, true);
    return resblock_part(c, "conv_l1", t, h, y1, B + 2, BRICK_H1, B + 4, BRICK_H1, true, true);
}

// After the encoder: the level-1 tensors, the brick's own planes of the level-1 input, and the part of conv_l1 that needs
// nothing from the neighbours -- it runs while the faces travel.
static int brick_interior(nbe_ctx* c, nbe_ctx::StreamState& st) {
    const int m = c->mid;
    const Tensor& td = st.td;
    const Layer* L1 = find_layer(c, "conv_l1", "conv_1");
    if (!L1) return fail("missing layer conv_l1/conv_1");
    st.t = tallocp(c, m, td.p.D + 2 * BRICK_H1, td.p.H, td.p.W, 1);
    if (st.t.off < 0) return fail("workspace exhausted (level 1 input)");
    st.h = alloc_hidden(c, m, st.t.p.D - 2, st.t, block_fused(c, L1, true));
    st.y1 = tallocp(c, m, st.t.p.D - 4, td.p.H, td.p.W, 1);
    if (st.h.off < 0 || st.y1.off < 0) return fail("workspace exhausted (level 1)");
    if (!c->dry) launch_wrap_pad(td.p, zview(st.t, BRICK_H1, td.p.D).p, 1, c->vel, c->stream, 0);
    return brick_conv_l1(c, st.t, st.h, st.y1, 0);
}

// With the neighbours' faces: the rest of conv_l1, the level-1 skip connection, down_l1 on the brick's own planes, and its
// boundary planes for the second exchange.
static int brick_edges(nbe_ctx* c, nbe_ctx::StreamState& st) {
    const int m = c->mid;
    Tensor& td = st.td;
    const int B = td.p.D;
    if (!c->dry) {
        launch_wrap_pad(brick_planes(c, td, c->bio.recv_lo, BRICK_H1), zview(st.t, 0, BRICK_H1).p, 1, c->vel, c->stream, 0);
        launch_wrap_pad(brick_planes(c, td, c->bio.recv_hi, BRICK_H1), zview(st.t, BRICK_H1 + B, BRICK_H1).p, 1, c->vel, c->stream, 0);
    }
    if (brick_conv_l1(c, st.t, st.h, st.y1, 1) || brick_conv_l1(c, st.t, st.h, st.y1, 2)) return 1;
    tfree(c, st.h); tfree(c, st.t); tfree(c, td);
    Tensor& y1 = st.y1;                                           // planes [-4, B + 4) of the brick's level-1 encoder output
    st.cat1 = tallocp(c, 2 * m, y1.p.D, y1.p.H - 2, y1.p.W - 2, 1);
    if (st.cat1.off < 0) return fail("workspace exhausted (cat1)");
    if (!c->dry) {
        Planes sp = y1.p; sp.G = c->mid / (c->prec == PREC_F16 ? 8 : 4);
        launch_crop(sp, 0, st.cat1.p, 0, c->vel, c->stream, 0);
    }
    st.t2 = talloc(c, m, B / 2, (y1.p.H - 2) / 2, (y1.p.W - 2) / 2);
    const Layer* Ld1 = find_layer(c, "down_l1", "conv_0");
    if (st.t2.off < 0 || !Ld1) return fail("workspace exhausted or missing layer (down_l1)");
    ConvLaunch cl; cl.in = inner(zview(y1, 4, B)); cl.Dv = st.t2.p.D; cl.Hv = st.t2.p.H; cl.Wv = st.t2.p.W; cl.out = st.t2.p; cl.flags = F_ACT;
    if (run_conv(c, *Ld1, cl, true)) return 1;
    tfree(c, y1);
    if (!c->dry && c->bio.send_lo) {
        launch_crop(zview(st.t2, 0, BRICK_H2).p, 0, brick_planes(c, st.t2, c->bio.send_lo, BRICK_H2), 0, c->vel, c->stream, 0);
        launch_crop(zview(st.t2, st.t2.p.D - BRICK_H2, BRICK_H2).p, 0, brick_planes(c, st.t2, c->bio.send_hi, BRICK_H2), 0, c->vel, c->stream, 0);
    }
    return 0;
}

// The level-2 input: the brick's own down_l1 planes between the neighbours' (second exchange), extended periodically by 10
// voxels in y and x.
static int brick_level2(nbe_ctx* c, nbe_ctx::StreamState& st, Tensor* t_out) {
    Tensor& t2 = st.t2;
    Tensor t = talloc(c, c->mid, t2.p.D + 2 * BRICK_H2, t2.p.H + 20, t2.p.W + 20);
    if (t.off < 0) return fail("workspace exhausted (level 2 input)");
    if (!c->dry) {
        launch_wrap_pad(t2.p, zview(t, BRICK_H2, t2.p.D).p, 10, c->vel, c->stream, 0);
        launch_wrap_pad(brick_planes(c, t2, c->bio.recv_lo, BRICK_H2), zview(t, 0, BRICK_H2).p, 10, c->vel, c->stream, 0);
        launch_wrap_pad(brick_planes(c, t2, c->bio.recv_hi, BRICK_H2), zview(t, BRICK_H2 + t2.p.D, BRICK_H2).p, 10, c->vel, c->stream, 0);
    }
    tfree(c, t2);
    *t_out = t;
    return 0;
}

// Everything from the level-2 input on: levels 2-3, the level-1 decoder, then the level-0 decoder slab by slab with the head.
static int stream_tail(nbe_ctx* c, const Tensor& tin, const HeadOut& ho, int S, Tensor skip0, Tensor cat1, Tensor t) {
    const int m = c->mid, pad = tin.pad;
    const int sy = pad ? 0 : 2;
    const Layer *Lr00 = find_layer(c, "conv_r00", "conv_1"), *Lr01 = find_layer(c, "conv_r01", "conv_1");
    if (!Lr00 || !Lr01) return fail("missing conv_1 layers of the level-0 blocks");
    Tensor y2, cat2, r;
    if (resblock(c, "conv_l2", t, true, true, m, m, &y2)) return 1;
    tfree(c, t);
    cat2 = talloc(c, 2 * m, y2.p.D - 8, y2.p.H - 8, y2.p.W - 8);
    if (cat2.off < 0) return fail("workspace exhausted (cat2)");
    crop_into(c, y2, 4, cat2);
    if (downblock(c, "down_l2", y2, &t)) return 1;
    tfree(c, y2);
    if (resblock(c, "conv_c", t, true, true, m, m, &r)) return 1;
    tfree(c, t);
    if (upblock(c, "up_r2", r, cat2)) return 1;
    tfree(c, r);
    if (resblock(c, "conv_r2", cat2, true, true, m, 2 * m, &r)) return 1;
    tfree(c, cat2);
    // periodic-yx: the level-2 result carries 2 voxels of y/x context that the periodic level 1 does not need
    if (upblock(c, "up_r1", r, cat1, pad ? 2 : 0)) return 1;
    fill_halo(c, cat1);
    tfree(c, r);
    if (resblock(c, "conv_r1", cat1, true, true, m, 2 * m, &r)) return 1;      // r: level-1 decoder output
    tfree(c, cat1);
    const int rcrop = 0;
    if (2 * r.p.D != skip0.p.D || 2 * (r.p.H - 2 * r.pad) != skip0.p.H - 2 * pad || 2 * (r.p.W - 2 * r.pad) != skip0.p.W - 2 * pad)
        return fail("internal: level-0 concat geometry mismatch");

    if (pad && c->zx && !c->dry && c->bio.skip_recv_lo) {
        // brick mode: the neighbours' planes of the skip connection, below and above the brick's own -- the last of the four
        // exchanges to be needed; it travelled while levels 1-3 ran, and only now does the stream wait for it
        if (c->bio.skip_ready) HIPCHK(hipStreamWaitEvent(c->stream, c->bio.skip_ready, 0));
        launch_crop(brick_planes(c, skip0, c->bio.skip_recv_lo, BRICK_H0), 0, zview(skip0, 0, BRICK_H0).p, 0, c->vel, c->stream, 0);
        launch_crop(brick_planes(c, skip0, c->bio.skip_recv_hi, BRICK_H0), 0, zview(skip0, skip0.p.D - BRICK_H0, BRICK_H0).p, 0, c->vel, c->stream, 0);
    }
    const int Yo = skip0.p.D - 8;                                 // output planes (= D - 96)
    // Persistent slab tensors of the level-0 decoder, with the same carry-over of the overlaps (8 / 6 / 4 / 2 planes of
    // the concat tensor, the hidden and the result of conv_r00, the hidden of conv_r01).
    const int Hs = skip0.p.H - 2 * pad, Ws = skip0.p.W - 2 * pad;
    // Fused blocks on the gauged f16x3 kernel read concat([skip, up]) from two tensors (core :168-169 without the concat):
    // the slab's planes of the skip connection where they are, the up-sampled half in a mid-channel tensor of its own.
    static const bool two_off = getenv("NBE_TWOSRC") && atoi(getenv("NBE_TWOSRC")) == 0;           // A/B switch
    const bool two = block_fused(c, Lr00, true) && !two_off && c->mid % 16 == 0;   // the kernel switches sources between 16-channel chunks
    Tensor cat = tallocp(c, two ? m : 2 * m, S + 8, Hs, Ws, pad), hq = alloc_hidden(c, 2 * m, S + 6, cat, block_fused(c, Lr00, true));
    Tensor q = tallocp(c, m, S + 4, Hs - 2 * sy, Ws - 2 * sy, pad), hy = alloc_hidden(c, m, S + 2, q, block_fused(c, Lr01, true));
    Tensor y = tallocp(c, c->out_chan, S, Hs - 4 * sy, Ws - 4 * sy, pad);
    if (cat.off < 0 || hq.off < 0 || q.off < 0 || hy.off < 0 || y.off < 0) return fail("workspace exhausted (level-0 decoder slabs)");
    for (int z = 0, n = 0; z < Yo; z += n) {
        n = std::min(S, Yo - z);
        if (c->pipe.active && c->pipe.out_async && Yo - z > PIPE_EDGE && Yo - z - n < PIPE_EDGE)
            n = Yo - z - PIPE_EDGE;                              // leave a short last slab (pipelined host path, see stream_encode)
        const bool first = z == 0;
        const int c0 = first ? 0 : 8, cn = first ? n + 8 : n;     // new planes of the concat tensor: [c0, c0 + cn)
        set_org(cat, skip0.org[0] + z, skip0.org[1], skip0.org[2]);  // slab-local plane j of the concat is plane z + j of the skip connection
        { int og[3]; org_conv(cat, 2, og); set_org(q, og[0], og[1], og[2]); }
        if (!two && !c->dry) launch_crop(zview(skip0, z + c0, cn).p, 0, zview(cat, c0, cn).p, 0, c->vel, c->stream, 0);
        if (upblock(c, "up_r0", zview(r, (z + c0) / 2, cn / 2), zview(cat, c0, cn), rcrop, two ? 0 : -1)) return 1;
        fill_halo(c, zview(cat, c0, cn));
        // two sources: slab-local plane j of the concat is plane z + j of the skip connection
        const Tensor sk = two ? zview(skip0, z, std::min(S + 8, skip0.p.D - z)) : cat;
        const Tensor* up2 = two ? &cat : nullptr;
        if (first) {
            if (resblock_part(c, "conv_r00", sk, hq, q, 0, n + 4, 0, n + 6, true, true, up2)) return 1;
            if (resblock_part(c, "conv_r01", q, hy, y, 0, n, 0, n + 2, true, false)) return 1;
        } else {
            if (resblock_part(c, "conv_r00", sk, hq, q, 4, n, 6, n, true, true, up2)) return 1;
            if (resblock_part(c, "conv_r01", q, hy, y, 0, n, 2, n, true, false)) return 1;
        }
        if (z + n < Yo) {
            carry_planes(c, cat, n, 0, 8);
            carry_planes(c, hq, n, 0, 6);
            carry_planes(c, q, n, 0, 4);
            carry_planes(c, hy, n, 0, 2);
        }
        run_head(c, zview(y, 0, n), zview(tin, z, n + 96), ho, z);
        if (c->pipe.active && c->pipe.out_async && !c->dry && pipe_output(c, z, n)) return 1;
        if (c->prog && !c->dry && z + n < Yo)                    // the tile's last slab is reported by the sub-box loop
            c->prog->post(c->pipe.active && c->pipe.out_async ? c->down_stream : c->stream,
                          c->prog_k * 1000 + (int)(1000L * (z + n) / Yo), c->prog_n * 1000);
    }
    tfree(c, cat); tfree(c, hq); tfree(c, q); tfree(c, hy); tfree(c, y);
    tfree(c, r); tfree(c, skip0);
    return 0;
}

static void stash_arena(nbe_ctx* c) { c->sst.blks = c->arena.blks; c->sst.high = c->arena.high; }

// phase 0: the whole schedule.  Brick mode (c->zx): 1 = encoder + faces of the down_l0 output, 2 = the interior of conv_l1,
// 3 = with the received faces up to the faces of the down_l1 output, 4 = with those, everything else.  The arena keeps the
// tensors in between (c->sst); any other use of the context drops them (sst.valid).
static int network_stream(nbe_ctx* c, const Tensor& tin, const HeadOut& ho, int S) {
    auto& st = c->sst;
    const int pad = tin.pad;
    const bool zx = pad && c->zx;
    if (c->phase >= 2) {
        if (!st.valid || st.stage != c->phase - 1) return fail("brick calls out of order (encode, interior, exchange, finish) or the context was used in between");
        c->arena.blks = st.blks; c->arena.high = st.high;
    }
    if (c->phase <= 1) {
        if (stream_encode(c, tin, ho, S, &st.skip0, &st.td)) return 1;
        st.tin = tin; st.S = S;
        if (c->phase == 1) {
            // the boundary planes of the down_l0 output for the neighbours
            if (!c->dry) {
                launch_crop(zview(st.td, 0, BRICK_H1).p, 0, brick_planes(c, st.td, c->bio.send_lo, BRICK_H1), 0, c->vel, c->stream, 0);
                launch_crop(zview(st.td, st.td.p.D - BRICK_H1, BRICK_H1).p, 0, brick_planes(c, st.td, c->bio.send_hi, BRICK_H1), 0, c->vel, c->stream, 0);
                // ... and the first / last four of the brick's own planes of the skip connection (planes 4 .. D - 4 of the tensor)
                const int own = st.skip0.p.D - 2 * BRICK_H0;
                launch_crop(zview(st.skip0, BRICK_H0, BRICK_H0).p, 0, brick_planes(c, st.skip0, c->bio.skip_send_lo, BRICK_H0), 0, c->vel, c->stream, 0);
                launch_crop(zview(st.skip0, own, BRICK_H0).p, 0, brick_planes(c, st.skip0, c->bio.skip_send_hi, BRICK_H0), 0, c->vel, c->stream, 0);
            }
            st.valid = true; st.stage = 1; stash_arena(c);
            return 0;
        }
    }
    if (!zx) {
        Tensor cat1, t;
        if (stream_level1(c, pad, pad && c->pz, st.td, &cat1, &t)) return 1;
        return stream_tail(c, tin, ho, S, st.skip0, cat1, t);
    }
    if (c->phase == 0 || c->phase == 2) {
        if (brick_interior(c, st)) return 1;
        if (c->phase == 2) { st.stage = 2; stash_arena(c); return 0; }
    }
    if (c->phase == 0 || c->phase == 3) {
        if (brick_edges(c, st)) return 1;
        if (c->phase == 3) { st.stage = 3; stash_arena(c); return 0; }
    }
    Tensor t;
    if (brick_level2(c, st, &t)) return 1;
    st.valid = false;
    return stream_tail(c, st.tin, ho, st.S, st.skip0, st.cat1, t);
}

// periodic-yx tiles: z as usual; y and x are the box itself (+ 2 halo voxels), a multiple of 8 with room for the
// 22 voxels of periodic context of the level-1 input
static int check_dims_pyx(int D, int H, int W) {
    if (D < 104 || D % 8 != 0) return fail("input depth %d unsupported: must be >= 104 and a multiple of 8", D);
    const int v[2] = {H - 2, W - 2};
    for (int i = 0; i < 2; ++i)
        if (v[i] < 48 || v[i] % 8 != 0) return fail("periodic extent %d unsupported: must be >= 48 and a multiple of 8", v[i]);
    return 0;
}

// bytes of workspace a (D,H,W) input needs with the current schedule (c->slab): a dry run of the network through
// the arena (no launches); < 0 on error
static int64_t workspace_need(nbe_ctx* c, int D, int H, int W) {
    c->dry = true;
    c->arena.reset();
    Tensor tin = talloc(c, c->in_chan, D, H, W), y;
    tin.pad = c->pyx ? 1 : 0;
    HeadOut ho{};
    const int rc = c->slab > 0 ? network_stream(c, tin, ho, c->slab) : network(c, tin, &y);
    c->dry = false;
    return rc ? -1 : c->arena.high;
}

// size the workspace for a (D,H,W) input with a dry run, then (re)allocate it
static int ensure_workspace(nbe_ctx* c, int D, int H, int W) {
    const int64_t need = workspace_need(c, D, H, W);
    if (need < 0) return 1;
    if (need > c->ws_bytes) {
        c->sst.valid = false;
        if (c->ws) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
        HIPCHK(hipMalloc((void**)&c->ws, need));
        // padded channel planes are read (against zero weights) but never written: they must hold finite values
        HIPCHK(hipMemsetAsync(c->ws, 0, need, c->stream));
        c->ws_bytes = need;
    }
    return 0;
}

static int require_ready(nbe_ctx* c) {
    if (!c->have_weights) return fail("No parameters loaded. Call nbe_load_style_weights / nbe_load_premod_weights first.");
    if (c->style && !c->modulated) return fail("style weights are loaded but nbe_set_cosmology(Om, Dz) has not been called");
    return 0;
}

// one sub-box: `box` is a device-resident (C, Db, Hb, Wb) volume, the crop origin may be negative (periodic)
static int run_subbox(nbe_ctx* c, const float* box, int Db, int Hb, int Wb, int o0, int o1, int o2,
                      int D, int H, int W, float Dz, float vel_fac, void* disp, void* velo, int out_dtype,
                      int OD, int OH, int OW, int a0, int a1, int a2) {
    c->sst.valid = false;                                        // a tile reuses the arena: a pending brick's tensors are gone
    c->arena.reset();
    Tensor tin = talloc(c, c->in_chan, D, H, W), y;
    tin.pad = c->pyx ? 1 : 0;                                   // periodic-yx: (H, W) = box extent + 2, gathered from origin - 1
    if (tin.pad) set_org(tin, 0, 48, 48);                        // its interior sits 48 voxels into the padded frame
    // core :132-134: x = x * (Dz / 6); the pipelined host path gathers slab by slab as the box arrives (pipe_input)
    if (!c->pipe.active && !c->pipe.slabwise)
        launch_gather(box, c->in_chan, Db, Hb, Wb, o0, o1, o2, tin.p, Dz / 6.0f * c->act_scale, c->prec, c->stream);
    const HeadOut ho{disp, velo, out_dtype, OD, OH, OW, a0, a1, a2, Dz, vel_fac};
    if (c->slab > 0) return network_stream(c, tin, ho, c->slab);
    if (network(c, tin, &y)) return 1;
    run_head(c, y, tin, ho, 0);
    return 0;
}

static void drop_graphs(nbe_ctx* c) {
    for (auto& kv : c->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    c->graphs.clear();
}

// One tile through run_subbox, replayed from a captured hipGraph when the identical tile (same pointers, geometry,
// scalars, weights epoch) has been run before.  The first request runs eagerly on the caller's stream (one-time
// hipFuncSetAttribute calls, lazily created state); the second is captured on the context's own stream -- the caller's
// may be the legacy null stream, which cannot be captured -- and every later one is a single hipGraphLaunch, fenced
// against the caller's stream by two events.  Not used with profiling, progress callbacks or the pipelined host path
// (they synchronise or use other streams inside the schedule).  NBE_GRAPH=0 turns it off.
static int run_tile(nbe_ctx* c, const float* box, int Db, int Hb, int Wb, int o0, int o1, int o2,
                    int D, int H, int W, float Dz, float vel_fac, void* disp, void* velo, int out_dtype,
                    int OD, int OH, int OW, int a0, int a1, int a2) {
    const bool off = getenv("NBE_GRAPH") && atoi(getenv("NBE_GRAPH")) == 0;
    if (off || c->prof || c->prog_cb || c->pipe.active || c->pipe.slabwise || c->dry || c->probe.on)
        return run_subbox(c, box, Db, Hb, Wb, o0, o1, o2, D, H, W, Dz, vel_fac, disp, velo, out_dtype, OD, OH, OW, a0, a1, a2);
    nbe_ctx::GraphKey k;
    memset(&k, 0, sizeof k);
    k.box = box; k.disp = disp; k.velo = velo; k.ws = c->ws;
    const int geo[18] = {Db, Hb, Wb, o0, o1, o2, D, H, W, out_dtype, OD, OH, OW, a0, a1, a2, c->slab, c->prec};
    memcpy(k.geo, geo, sizeof geo);
    k.f[0] = Dz; k.f[1] = vel_fac; k.f[2] = c->act_scale;
    // (the A/B switches that launchers read per launch are part of the key: a captured graph holds the kernels they chose)
    auto sw = [](const char* n, int bit) { const char* e = getenv(n); return (e && atoi(e) == 0) ? (1 << bit) : 0; };
    k.epoch = c->epoch; k.flags = (c->pyx ? 1 : 0) | (c->pz ? 2 : 0) | (c->gauge_active ? 4 : 0) | (c->fuse ? 8 : 0) |
              sw("NBE_WINO", 4) | sw("NBE_UP8", 5) | sw("NBE_STEM", 6) | sw("NBE_H3G_TALL", 7) | sw("NBE_NARROW", 8) | sw("NBE_HEAD4", 9);
    nbe_ctx::GraphVal& g = c->graphs[k];
    g.used = ++c->graph_clock;
    if (!g.exec && g.seen++ == 0) {                              // first time: eager
        if (c->graphs.size() > 16) {                             // keep the cache small: drop the least recently used
            auto lru = c->graphs.begin();
            for (auto it = c->graphs.begin(); it != c->graphs.end(); ++it) if (it->second.used < lru->second.used) lru = it;
            if (lru->second.exec) (void)hipGraphExecDestroy(lru->second.exec);
            if (lru->second.graph) (void)hipGraphDestroy(lru->second.graph);
            c->graphs.erase(lru);
        }
        return run_subbox(c, box, Db, Hb, Wb, o0, o1, o2, D, H, W, Dz, vel_fac, disp, velo, out_dtype, OD, OH, OW, a0, a1, a2);
    }
    if (!c->ev_g0) { HIPCHK(hipEventCreateWithFlags(&c->ev_g0, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_g1, hipEventDisableTiming)); }
    hipStream_t user = c->stream;
    if (!g.exec) {                                               // second time: capture on the own stream
        c->stream = c->own_stream;
        hipError_t e = hipStreamBeginCapture(c->own_stream, hipStreamCaptureModeThreadLocal);
        int rc = 0;
        if (e == hipSuccess) {
            rc = run_subbox(c, box, Db, Hb, Wb, o0, o1, o2, D, H, W, Dz, vel_fac, disp, velo, out_dtype, OD, OH, OW, a0, a1, a2);
            e = hipStreamEndCapture(c->own_stream, &g.graph);
            if (e == hipSuccess && !rc) e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
        }
        c->stream = user;
        if (rc || e != hipSuccess || !g.exec) {                  // capture is an optimisation: fall back to eager for good
            (void)hipGetLastError();
            if (g.graph) { (void)hipGraphDestroy(g.graph); g.graph = nullptr; }
            g.exec = nullptr; g.seen = -1000000;
            return run_subbox(c, box, Db, Hb, Wb, o0, o1, o2, D, H, W, Dz, vel_fac, disp, velo, out_dtype, OD, OH, OW, a0, a1, a2);
        }
    }
    if (user != c->own_stream) { HIPCHK(hipEventRecord(c->ev_g0, user)); HIPCHK(hipStreamWaitEvent(c->own_stream, c->ev_g0, 0)); }
    HIPCHK(hipGraphLaunch(g.exec, c->own_stream));
    if (user != c->own_stream) { HIPCHK(hipEventRecord(c->ev_g1, c->own_stream)); HIPCHK(hipStreamWaitEvent(user, c->ev_g1, 0)); }
    ++c->graph_replays;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
static void free_layers(nbe_ctx* c) {
    drop_graphs(c); ++c->epoch;
    for (auto& kv : c->layers) {
        Layer& L = kv.second;
        (void)hipFree(L.weight); (void)hipFree(L.sw); (void)hipFree(L.sb); (void)hipFree(L.wn); (void)hipFree(L.dwn);
        (void)hipFree(L.pw.w); (void)hipFree(L.pw.dw); (void)hipFree(L.pw.bias); (void)hipFree(L.bias0); (void)hipFree(L.pw.stem); (void)hipFree(L.pw.ww); (void)hipFree(L.pwn.w); (void)hipFree(L.pwn.dw); (void)hipFree(L.bias_f);
        (void)hipFree(L.alpha); (void)hipFree(L.beta); (void)hipFree(L.dwn_f);
    }
    c->layers.clear();
    c->bias_scale = 1.f; c->bias_max = 0.f; c->bias_dirty = true;
    c->have_weights = false; c->modulated = false; c->gauge = false; c->gauge_active = false; c->fuse = false; c->novel_fuse = false; c->wino_ok = false;
}

static int kind_of(const nbe_layer_desc& d, int* kind) {
    const std::string blk = d.block, lay = d.layer;
    if (lay == "skip") { if (d.k != 1) return fail("%s/%s: skip layers have k=1", d.block, d.layer); *kind = 1; return 0; }
    if (blk.rfind("down_", 0) == 0) { if (d.k != 2) return fail("%s: down layers have k=2", d.block); *kind = 2; return 0; }
    if (blk.rfind("up_", 0) == 0) { if (d.k != 2) return fail("%s: up layers have k=2", d.block); *kind = 3; return 0; }
    if (d.k != 3) return fail("%s/%s: conv layers have k=3", d.block, d.layer);
    *kind = 0;
    return 0;
}

static int expected_shape(nbe_ctx* c, const std::string& blk, const std::string& lay, int* cout, int* cin) {
    const int m = c->mid;
    int bi, bo;
    if (blk == "conv_l00") { bi = c->in_chan; bo = m; }
    else if (blk == "conv_r2" || blk == "conv_r1" || blk == "conv_r00") { bi = 2 * m; bo = m; }
    else if (blk == "conv_r01") { bi = m; bo = c->out_chan; }
    else { bi = m; bo = m; }
    const int midc = bi > bo ? bi : bo;                     // style_blocks_vel.py:126
    if (lay == "skip") { *cin = bi; *cout = bo; }
    else if (blk.rfind("down_", 0) == 0 || blk.rfind("up_", 0) == 0) { *cin = bi; *cout = bo; }
    else if (lay == "conv_0") { *cin = bi; *cout = midc; }
    else if (lay == "conv_1") { *cin = midc; *cout = bo; }
    else return fail("unknown layer %s/%s", blk.c_str(), lay.c_str());
    return 0;
}

static const char* kBlocks[15] = {"conv_l00", "conv_l01", "down_l0", "conv_l1", "down_l1", "conv_l2", "down_l2", "conv_c",
                                  "up_r2", "conv_r2", "up_r1", "conv_r1", "up_r0", "conv_r00", "conv_r01"};

// Winograd-z weights of the gauged wide 3x3x3 layers (conv_h3w_kernel), from the modulated weights L.wn that are current
static int pack_wino(nbe_ctx* c) {
    c->wino_ok = false;
    if (!((c->prec == PREC_F16X3 && (c->vel ? c->gauge_active : true)) || (c->prec == PREC_F16 && c->vel && c->gauge_active))) return 0;
    if (!c->wino_flag) HIPCHK(hipMalloc((void**)&c->wino_flag, 4));
    HIPCHK(hipMemsetAsync(c->wino_flag, 0, 4, c->stream));
    for (auto& kv : c->layers) {
        Layer& L = kv.second;
        if (L.pw.ww && L.kind == 0 && (L.g6 || !c->vel)) launch_pack_h3w(L.wn, L.cout, L.cin, L.pw.cin_pad, L.pw.ctiles, L.pw.ww, c->wino_flag, c->stream, c->prec);
        if (L.pw.ww && L.kind == 1 && (c->vel ? (L.b_sub && c->fuse) : c->novel_fuse)) {   // a fused skip: [W_s | dW_s~] for conv_h3w_kernel<SKIP>
            launch_pack_h3w_skip(L.wn, L.cout, L.cin, L.pw, L.pw.ww, c->wino_flag, c->stream);
            if (c->vel) launch_pack_h3w_skip(c->prec == PREC_F16 ? L.dwn_f : L.dwn, L.cout, L.cin, L.pw, L.pw.ww + L.pw.floats, c->wino_flag, c->stream);
        }
    }
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, c->wino_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->wino_ok = bad == 0;
    if (!c->vel) c->fuse = c->novel_fuse && c->wino_ok;         // displacement only: the fused skips live in the Winograd-z kernel
    if (c->prec == PREC_F16) c->fuse = c->fuse && c->wino_ok;   // the float16 model: likewise
    return 0;
}

// Displacement-only f16x3 networks: every block whose conv_1 has a Winograd-z form runs its 1x1x1 skip inside that launch
// (conv_h3w_kernel<SKIP, NOVEL>), as the velocity networks do through wire_gauge
static int wire_novel(nbe_ctx* c) {
    c->novel_fuse = false;
    static const bool no_fuse = getenv("NBE_FUSE") && atoi(getenv("NBE_FUSE")) == 0;            // A/B switch
    if (c->vel || c->prec != PREC_F16X3 || no_fuse) return 0;
    for (const char* b : kBlocks) {
        if (!strncmp(b, "down_", 5) || !strncmp(b, "up_", 3)) continue;
        auto i1 = c->layers.find(std::string(b) + "/conv_1"), is = c->layers.find(std::string(b) + "/skip");
        if (i1 == c->layers.end() || is == c->layers.end()) return fail("internal: block %s", b);
        Layer &L1 = i1->second, &Ls = is->second;
        if (!L1.pw.ww || !Ls.pw.ww || 2 * (Ls.pw.cin_pad / 16) > 16 || L1.pw.ctiles != Ls.pw.ctiles) return 0;   // all blocks or none
    }
    for (const char* b : kBlocks) {
        if (!strncmp(b, "down_", 5) || !strncmp(b, "up_", 3)) continue;
        Layer &L1 = c->layers[std::string(b) + "/conv_1"], &Ls = c->layers[std::string(b) + "/skip"];
        L1.fskip = &Ls;
        const int nb = L1.pw.ctiles * 32 * L1.pw.ni;
        HIPCHK(hipMalloc((void**)&L1.bias_f, nb * 4));
    }
    c->novel_fuse = true;
    return 0;
}

// Tangent gauges of the style path (conv_h3g_kernel): every tensor with a tangent stores dx + a (.) x, a = the alpha
// of the one 3x3x3 layer that reads it, so that layer runs two products instead of three; the tensor's other readers
// (skips, down-sampling) fold a into their tangent weights.  Tensors read only by general kernels keep a = 0.
static int wire_gauge(nbe_ctx* c) {
    const int m = c->mid;
    for (auto& kv : c->layers) {
        Layer& L = kv.second;
        const size_t na = (size_t)roundup(L.cin, 16) + 64, nb = (size_t)L.pw.ctiles * 32 * L.pw.ni + 64;
        HIPCHK(hipMalloc((void**)&L.alpha, na * 4)); HIPCHK(hipMemset(L.alpha, 0, na * 4));
        HIPCHK(hipMalloc((void**)&L.beta, nb * 4)); HIPCHK(hipMemset(L.beta, 0, nb * 4));
    }
    if (!c->gauge_flag) HIPCHK(hipMalloc((void**)&c->gauge_flag, 4));
    auto lay = [&](const char* b, const char* l) -> Layer* {
        auto it = c->layers.find(std::string(b) + "/" + l);
        return it == c->layers.end() ? nullptr : &it->second;
    };
    // output of `producer` is read by the 3x3x3 layer `consumer`/conv_0 as its input channels [off, off + cout)
    struct Rule { const char* pb; const char* pl; const char* consumer; int off; };
    const Rule rules[] = {
        {"conv_l00", "conv_1", "conv_l01", 0}, {"conv_l01", "conv_1", "conv_r00", 0}, {"down_l0", "conv_0", "conv_l1", 0},
        {"conv_l1", "conv_1", "conv_r1", 0},   {"down_l1", "conv_0", "conv_l2", 0},   {"conv_l2", "conv_1", "conv_r2", 0},
        {"down_l2", "conv_0", "conv_c", 0},    {"up_r2", "conv_0", "conv_r2", m},     {"up_r1", "conv_0", "conv_r1", m},
        {"up_r0", "conv_0", "conv_r00", m},    {"conv_r00", "conv_1", "conv_r01", 0},
    };
    for (const Rule& r : rules) {
        Layer *P = lay(r.pb, r.pl), *C = lay(r.consumer, "conv_0");
        if (!P || !C || r.off + P->cout > C->cin) return fail("internal: gauge wiring %s/%s -> %s", r.pb, r.pl, r.consumer);
        P->gout = C->alpha + r.off;
    }
    for (const char* b : kBlocks) {
        if (!strncmp(b, "down_", 5) || !strncmp(b, "up_", 3)) continue;
        Layer *L0 = lay(b, "conv_0"), *L1 = lay(b, "conv_1"), *Ls = lay(b, "skip");
        L0->gout = L1->alpha;                                    // the hidden tensor is read by conv_1 only
        L1->g6 = true;
        if (strcmp(b, "conv_l00")) { L0->g6 = true; Ls->a_in = L0->alpha; }   // conv_l00 reads the input field: no tangent
        // the skip can run inside conv_1 (conv_h3g_kernel<false>): f16x3, the block input has a tangent, the wide tile,
        // and the groups of both fit the kernel's table
        static const bool no_fuse = getenv("NBE_FUSE") && atoi(getenv("NBE_FUSE")) == 0;        // A/B switch
        if (c->prec == PREC_F16X3 && !no_fuse && (!L1->pwn.w || Ls->pwn.dw) &&
            3 * (L1->pw.cin_pad / 16) + Ls->pw.cin_pad / 16 <= NBE_MAX_GROUPS) {
            L1->fskip = Ls; Ls->b_sub = L1->beta;
            const int nb = L1->pw.ctiles * 32 * L1->pw.ni;
            HIPCHK(hipMalloc((void**)&L1->bias_f, nb * 4));
        }
        // float16 model (style path): the skip runs inside conv_h3w_kernel<SKIP, ., F16> wherever conv_1's launch has that form
        if (c->prec == PREC_F16 && !no_fuse && L1->pw.ww && Ls->pw.ww && Ls->dwn_f && L1->pw.ctiles == Ls->pw.ctiles &&
            2 * (Ls->pw.cin_pad / 32) <= 16) {                   // NBE_MAX_WSKIP (nbe_kernels_wino.h)
            L1->fskip = Ls; Ls->b_sub = L1->beta;
            const int nb = L1->pw.ctiles * 32 * L1->pw.ni;
            HIPCHK(hipMalloc((void**)&L1->bias_f, nb * 4));
        }
    }
    // every gauged 3x3x3 layer must fit the group table of conv_h3g_kernel
    for (auto& kv : c->layers)
        if (kv.second.g6 && 3 * (kv.second.pw.cin_pad / 16) > NBE_MAX_GROUPS) { c->gauge = false; return 0; }
    lay("down_l0", "conv_0")->a_in = lay("conv_r00", "conv_0")->alpha;    // they read conv_l01 / conv_l1 / conv_l2's output
    lay("down_l1", "conv_0")->a_in = lay("conv_r1", "conv_0")->alpha;
    lay("down_l2", "conv_0")->a_in = lay("conv_r2", "conv_0")->alpha;
    c->gauge = true;
    return 0;
}

// Premodulated (W, dW) pairs: modulate_emulator_parameters_vel (nbody_emulator.py:221-266) produces dW = W (.) (alpha[ci] +
// beta[co]) -- recognise that from the numbers (weighted alternating least squares for the additive model, then an
// element-wise check) and run the gauged kernels; any 3x3x3 layer whose pair does not factorise to float32 rounding
// (hand-made or perturbed dweight) leaves the whole network on the general three-product kernels.
static int wire_gauge_premod(nbe_ctx* c, const nbe_layer_desc* descs, int n) {
    std::map<std::string, std::vector<double>> al, be;
    std::map<std::string, const nbe_layer_desc*> by_name;
    for (int i = 0; i < n; ++i) by_name[std::string(descs[i].block) + "/" + descs[i].layer] = &descs[i];
    for (const char* b : kBlocks) {
        if (!strncmp(b, "down_", 5) || !strncmp(b, "up_", 3)) continue;
        for (const char* l : {"conv_0", "conv_1"}) {
            if (!strcmp(b, "conv_l00") && !strcmp(l, "conv_0")) continue;    // reads the input field: never gauged
            const std::string key = std::string(b) + "/" + l;
            const nbe_layer_desc& d = *by_name[key];
            const int co = d.cout, ci = d.cin, k3 = d.k * d.k * d.k;
            std::vector<double> a(ci, 0.0), bt(co, 0.0);
            double dmax = 0.0;
            for (size_t e = 0; e < (size_t)co * ci * k3; ++e) dmax = std::max(dmax, (double)std::fabs(d.dweight[e]));
            for (int iter = 0; iter < 200; ++iter) {
                double change = 0.0;
                for (int i = 0; i < ci; ++i) {                           // alpha[i] = sum w (dW - W beta) / sum w^2 over (o, k)
                    double num = 0.0, den = 0.0;
                    for (int o = 0; o < co; ++o)
                        for (int k = 0; k < k3; ++k) {
                            const double w = d.weight[((size_t)o * ci + i) * k3 + k], dw = d.dweight[((size_t)o * ci + i) * k3 + k];
                            num += w * (dw - w * bt[o]); den += w * w;
                        }
                    const double v = den > 0 ? num / den : 0.0;
                    change = std::max(change, std::fabs(v - a[i])); a[i] = v;
                }
                for (int o = 0; o < co; ++o) {
                    double num = 0.0, den = 0.0;
                    for (int i = 0; i < ci; ++i)
                        for (int k = 0; k < k3; ++k) {
                            const double w = d.weight[((size_t)o * ci + i) * k3 + k], dw = d.dweight[((size_t)o * ci + i) * k3 + k];
                            num += w * (dw - w * a[i]); den += w * w;
                        }
                    const double v = den > 0 ? num / den : 0.0;
                    change = std::max(change, std::fabs(v - bt[o])); bt[o] = v;
                }
                if (change < 1e-13) break;
            }
            double res = 0.0;
            for (int o = 0; o < co; ++o)
                for (int i = 0; i < ci; ++i)
                    for (int k = 0; k < k3; ++k) {
                        const size_t e = ((size_t)o * ci + i) * k3 + k;
                        res = std::max(res, std::fabs((double)d.dweight[e] - (double)d.weight[e] * (a[i] + bt[o])));
                    }
            if (!(res <= 2e-6 * dmax + 1e-30)) return 0;                 // does not factorise: keep the general kernels
            // alpha + c, beta - c is the same pair: centre alpha, and keep it small (the f16 formats store dx + alpha * x)
            const double amin = *std::min_element(a.begin(), a.end()), amax = *std::max_element(a.begin(), a.end());
            const double mid = 0.5 * (amin + amax);
            for (double& v : a) v -= mid;
            for (double& v : bt) v += mid;
            if (amax - mid > 64.0) return 0;
            al[key] = a; be[key] = bt;
        }
    }
    if (wire_gauge(c)) return 1;
    if (!c->gauge) return 0;                                    // a layer too wide for the gauged kernel's group table
    for (auto& kv : c->layers) {
        Layer& L = kv.second;
        auto ia = al.find(kv.first);
        if (ia != al.end()) {
            std::vector<float> fa(ia->second.begin(), ia->second.end()), fb(be[kv.first].begin(), be[kv.first].end());
            HIPCHK(hipMemcpy(L.alpha, fa.data(), fa.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(L.beta, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
        }
    }
    // general layers that read a gauged tensor: dW - W (.) a_in, a_in = alpha of the tensor's 3x3x3 reader (wire_gauge)
    auto fold = [&](const char* b, const char* l, const char* reader, int off) -> int {
        const std::string key = std::string(b) + "/" + l;
        const nbe_layer_desc& d = *by_name[key];
        Layer& L = c->layers[key];
        static const std::vector<double> none(4096, 0.0);          // reader == nullptr: the input carries no gauge (conv_l00)
        const std::vector<double>& a = reader ? al[std::string(reader) + "/conv_0"] : none;
        // a skip that runs inside its block's conv_1 (Layer::b_sub): the kernel's epilogue adds beta_1[o] * (W_s.x) as well
        const std::vector<double>* bsub = L.b_sub ? &be[std::string(b) + "/conv_1"] : nullptr;
        const int k3 = d.k * d.k * d.k;
        std::vector<float> eff((size_t)d.cout * d.cin * k3);
        for (int o = 0; o < d.cout; ++o)
            for (int i = 0; i < d.cin; ++i)
                for (int k = 0; k < k3; ++k) {
                    const size_t e = ((size_t)o * d.cin + i) * k3 + k;
                    eff[e] = (float)((double)d.dweight[e] - (double)d.weight[e] * a[off + i]
                                     - (bsub ? (double)d.weight[e] * (*bsub)[o] : 0.0));
                }
        HIPCHK(hipMemcpy(L.dwn, eff.data(), eff.size() * 4, hipMemcpyHostToDevice));
        launch_pack(L.dwn, d.cout, d.cin, L.kind, L.pw, L.pw.dw, c->stream);
        if (L.pwn.dw) launch_pack(L.dwn, d.cout, d.cin, L.kind, L.pwn, L.pwn.dw, c->stream);
        return 0;
    };
    for (const char* b : kBlocks) {
        if (!strncmp(b, "down_", 5) || !strncmp(b, "up_", 3)) continue;
        if (!strcmp(b, "conv_l00")) { if (c->layers[std::string(b) + "/skip"].b_sub && fold(b, "skip", nullptr, 0)) return 1; continue; }
        if (fold(b, "skip", b, 0)) return 1;
    }
    if (fold("down_l0", "conv_0", "conv_r00", 0) || fold("down_l1", "conv_0", "conv_r1", 0) ||
        fold("down_l2", "conv_0", "conv_r2", 0)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->gauge_active = c->gauge;
    c->fuse = c->gauge && c->prec == PREC_F16X3;
    return pack_wino(c);
}

static int load_weights(nbe_ctx* c, const nbe_layer_desc* descs, int n, bool style) {
    c->sst.valid = false;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_layers(c);
    for (int i = 0; i < n; ++i) {
        const nbe_layer_desc& d = descs[i];
        if (!d.block || !d.layer || !d.weight || !d.bias) return fail("layer %d: block, layer, weight and bias are required", i);
        Layer L;
        L.block = d.block; L.layer = d.layer; L.cout = d.cout; L.cin = d.cin; L.k = d.k;
        if (kind_of(d, &L.kind)) return 1;
        int ec, ei;
        if (expected_shape(c, L.block, L.layer, &ec, &ei)) return 1;
        if (ec != d.cout || ei != d.cin)
            return fail("%s/%s: weight shape (%d,%d,k) does not match the architecture (%d,%d,k)", d.block, d.layer, d.cout, d.cin, ec, ei);
        L.first = (L.block == "conv_l00") && (L.layer == "conv_0" || L.layer == "skip");     // nbody_emulator.py:243-246
        const size_t nw = (size_t)d.cout * d.cin * d.k * d.k * d.k;
        PackedW& pw = L.pw;
        pw.mode = L.kind == 0 ? MODE_FLAT3 : (L.kind == 2 ? MODE_DOWN : MODE_FLAT1);
        pw.prec = c->prec;
        pw.ni = (prec_is_half(c->prec) || d.cout > 32) ? 2 : 1;
        pw.cin = d.cin; pw.cout = d.cout;
        pw.cin_pad = roundup(d.cin, prec_ck(c->prec, pw.mode));
        pw.ctiles = (d.cout + 32 * pw.ni - 1) / (32 * pw.ni);
        pw.nsets = L.kind == 3 ? 8 : 1;
        pw.floats = (int64_t)pw.ctiles * 32 * pw.ni * mode_nseg(pw.mode) * mode_taps(pw.mode) * pw.cin_pad / (c->prec == PREC_F16 ? 2 : 1);
        HIPCHK(hipMalloc((void**)&pw.w, pw.floats * pw.nsets * 4));
        if (c->vel) HIPCHK(hipMalloc((void**)&pw.dw, pw.floats * pw.nsets * 4));
        const int nb = pw.ctiles * 32 * pw.ni;
        HIPCHK(hipMalloc((void**)&pw.bias, nb * 4));
        HIPCHK(hipMemset(pw.bias, 0, nb * 4));
        HIPCHK(hipMemcpy(pw.bias, d.bias, d.cout * 4, hipMemcpyHostToDevice));
        // (cout <= 4: the head convolution 64 -> 3.  Narrow test models, cout 8 or 16, stay on the wide tile so that they
        // exercise what production-width layers run, skip fusion included.)
        if (c->prec == PREC_F16X3 && c->vel && (L.kind == 0 || L.kind == 1) && d.cout <= 4 && !L.first) {
            PackedW& pn = L.pwn;                               // same layer, 16-cout tiles (conv_h3g_kernel<true>)
            pn = pw; pn.w = nullptr; pn.dw = nullptr;
            pn.cout_t = 16; pn.ctiles = 1;
            pn.floats = (int64_t)16 * mode_nseg(pn.mode) * mode_taps(pn.mode) * pn.cin_pad;
            HIPCHK(hipMalloc((void**)&pn.w, pn.floats * 4));
            if (L.kind == 1) HIPCHK(hipMalloc((void**)&pn.dw, pn.floats * 4));   // a skip that runs inside the narrow conv_1
        }
        // Winograd-z packing (conv_h3w_kernel): 4 transformed kernels per 3 dz slices, wide tile only, Cin <= 128
        if ((c->prec == PREC_F16X3 && L.kind == 0 && !L.first && !L.pwn.w && pw.cin_pad / 16 <= 8) ||
            (L.kind == 0 && !L.first && wino_f16_layer(c->prec, c->vel, pw.cin_pad)))
            HIPCHK(hipMalloc((void**)&pw.ww, pw.floats * 4 / 3 * 4));
        if (c->prec == PREC_F16X3 && L.kind == 1 && !L.pwn.w && pw.cin_pad / 16 <= 8)     // a skip that may run fused: W_s and dW_s~
            HIPCHK(hipMalloc((void**)&pw.ww, pw.floats * 2 * 4));
        if (style && L.kind == 1 && !L.first && wino_f16_layer(c->prec, c->vel, pw.cin_pad)) {   // float16 model, style path: the same
            HIPCHK(hipMalloc((void**)&pw.ww, pw.floats * 2 * 4));
            HIPCHK(hipMalloc((void**)&L.dwn_f, nw * 4));
        }
        // the first layer in its own packing (stem_h3_kernel): K = 27 taps x 3 channels = 81 <= 96
        if (prec_is_half(c->prec) && L.kind == 0 && L.first && d.cin <= 3 && d.cout <= 64)
            HIPCHK(hipMalloc((void**)&pw.stem, 4 * 3 * 4 * 64 * 16));
        HIPCHK(hipMalloc((void**)&L.bias0, nb * 4));
        HIPCHK(hipMemcpy(L.bias0, pw.bias, nb * 4, hipMemcpyDeviceToDevice));
        for (int i = 0; i < d.cout; ++i)
            if (std::isfinite(d.bias[i])) c->bias_max = std::max(c->bias_max, std::fabs(d.bias[i]));
        HIPCHK(hipMalloc((void**)&L.wn, nw * 4));
        if (c->vel) HIPCHK(hipMalloc((void**)&L.dwn, nw * 4));
        if (style) {
            if (!d.style_weight || !d.style_bias) return fail("%s/%s: style_weight and style_bias are required", d.block, d.layer);
            HIPCHK(hipMalloc((void**)&L.weight, nw * 4));
            HIPCHK(hipMalloc((void**)&L.sw, d.cin * 2 * 4));
            HIPCHK(hipMalloc((void**)&L.sb, d.cin * 4));
            HIPCHK(hipMemcpy(L.weight, d.weight, nw * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(L.sw, d.style_weight, d.cin * 2 * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(L.sb, d.style_bias, d.cin * 4, hipMemcpyHostToDevice));
        } else {
            if (c->vel && !d.dweight) return fail("%s/%s: dweight is required for premodulated velocity weights", d.block, d.layer);
            HIPCHK(hipMemcpy(L.wn, d.weight, nw * 4, hipMemcpyHostToDevice));
            if (c->vel) HIPCHK(hipMemcpy(L.dwn, d.dweight, nw * 4, hipMemcpyHostToDevice));
            launch_pack(L.wn, d.cout, d.cin, L.kind, pw, pw.w, c->stream);
            if (c->vel) launch_pack(L.dwn, d.cout, d.cin, L.kind, pw, pw.dw, c->stream);
            if (L.pwn.w) launch_pack(L.wn, d.cout, d.cin, L.kind, L.pwn, L.pwn.w, c->stream);
            if (L.pwn.dw) launch_pack(L.dwn, d.cout, d.cin, L.kind, L.pwn, L.pwn.dw, c->stream);
        }
        c->layers[L.block + "/" + L.layer] = L;
    }
    // completeness: 9 ResNet blocks x {skip, conv_0, conv_1} + 6 resample blocks x {conv_0} = 33 layers
    for (const char* b : kBlocks) {
        const bool rs = !strncmp(b, "down_", 5) || !strncmp(b, "up_", 3);
        const char* need[3] = {"conv_0", rs ? nullptr : "skip", rs ? nullptr : "conv_1"};
        for (const char* l : need)
            if (l && !find_layer(c, b, l)) return fail("parameter tree is missing %s/%s", b, l);
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_weights = true; c->style = style; c->modulated = !style;
    c->mod_Om = NAN; c->mod_Dz = NAN;
    const char* ge = getenv("NBE_GAUGE");
    if (c->vel && !(ge && atoi(ge) == 0)) return style ? wire_gauge(c) : wire_gauge_premod(c, descs, n);
    if (!c->vel) {
        if (wire_novel(c)) return 1;
        if (!style) return pack_wino(c);                         // premodulated weights are final: pack their Winograd-z form now
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* nbe_last_error(void) { return g_err.c_str(); }
int nbe_version(void) { return 100; }

int nbe_create(int device_id, nbe_ctx** out) {
    if (!out) return fail("nbe_create: out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("nbe_create: no HIP device is visible; this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("nbe_create: device %d out of range (0..%d)", device_id, ndev - 1);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail("nbe_create: device %d is %s; the kernels are built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
    nbe_ctx* c = new nbe_ctx();
    c->device = device_id;
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    if (const char* e = getenv("NBE_MAX_TILE")) c->max_tile = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("NBE_SLAB")) c->slab_forced = atoi(e) >= 0 ? (atoi(e) & ~1) : -1;
    if (const char* e = getenv("NBE_PERIODIC")) c->pyx_allowed = atoi(e) != 0;
    *out = c;
    return 0;
}

int nbe_destroy(nbe_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    free_layers(c);
    (void)hipFree(c->probe.bits); (void)hipFree(c->probe.count);
    (void)hipFree(c->ws); (void)hipFree(c->box_in); (void)hipFree(c->box_out); (void)hipFree(c->gauge_flag); (void)hipFree(c->wino_flag); (void)hipFree(c->flags);
    drop_graphs(c);
    if (c->ev_g0) (void)hipEventDestroy(c->ev_g0);
    if (c->ev_g1) (void)hipEventDestroy(c->ev_g1);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < nbe_ctx::NSTAGE; ++i) { if (c->stage_buf[i]) (void)hipHostFree(c->stage_buf[i]); if (c->stage_free[i]) (void)hipEventDestroy(c->stage_free[i]); }
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->ev_down) (void)hipEventDestroy(c->ev_down);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

int nbe_set_stream(nbe_ctx* c, void* s) {
    if (!c) return fail("null context");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)s;      // NULL is the device's default (null) stream, as everywhere in HIP
    return 0;
}

int nbe_use_own_stream(nbe_ctx* c) {
    if (!c) return fail("null context");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return 0;
}

int nbe_synchronize(nbe_ctx* c) {
    if (!c) return fail("null context");
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int nbe_set_arch(nbe_ctx* c, int in_chan, int out_chan, int mid_chan, float eps, int compute_vel) {
    if (!c) return fail("null context");
    if (in_chan < 1 || in_chan > 16) return fail("in_chan=%d unsupported (1..16)", in_chan);
    if (out_chan < 1 || out_chan > 64) return fail("out_chan=%d unsupported (1..64)", out_chan);
    if (out_chan != in_chan) return fail("out_chan (%d) must equal in_chan (%d): the head adds the cropped input (core :187)", out_chan, in_chan);
    if (mid_chan < 8 || mid_chan % 8 != 0) return fail("mid_chan=%d unsupported (multiple of 8 required)", mid_chan);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_layers(c);
    c->in_chan = in_chan; c->out_chan = out_chan; c->mid = mid_chan; c->eps = eps; c->vel = compute_vel != 0;
    return 0;
}

int nbe_load_style_weights(nbe_ctx* c, const nbe_layer_desc* layers, int n) {
    if (!c || !layers) return fail("null argument");
    return load_weights(c, layers, n, true);
}

int nbe_load_premod_weights(nbe_ctx* c, const nbe_layer_desc* layers, int n) {
    if (!c || !layers) return fail("null argument");
    return load_weights(c, layers, n, false);
}

int nbe_set_cosmology(nbe_ctx* c, float Om, float Dz) {
    if (!c) return fail("null context");
    if (!c->have_weights) return fail("No parameters loaded. Call nbe_load_style_weights first.");
    if (!c->style) return 0;
    if (c->modulated && c->mod_Om == Om && c->mod_Dz == Dz) return 0;
    c->sst.valid = false;                                        // a pending brick was encoded with the previous modulation
    HIPCHK(hipSetDevice(c->device));
    // s = ((Om - 0.3) * 5, Dz - 1) in float32 (core :126-128)
    const float s0 = (Om - 0.3f) * 5.0f, s1 = Dz - 1.0f;
    bool use_gauge = c->gauge;
    if (use_gauge) {
        // alpha of every layer first (the tangent weights of the general layers need their neighbours'); a style
        // factor that is zero at this cosmology has no alpha: fall back to the three-product kernels for this call
        HIPCHK(hipMemsetAsync(c->gauge_flag, 0, 4, c->stream));
        for (auto& kv : c->layers)
            launch_style_alpha(kv.second.sw, kv.second.sb, kv.second.cin, s0, s1, kv.second.alpha, c->gauge_flag, c->stream);
        int bad = 0;
        HIPCHK(hipMemcpyAsync(&bad, c->gauge_flag, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        use_gauge = bad == 0;
    }
    for (auto& kv : c->layers) {
        Layer& L = kv.second;
        // (map order: a block's conv_1, which writes its beta, comes before its skip, which folds it in as b_sub)
        launch_modulate(L.weight, L.sw, L.sb, L.cout, L.cin, L.k * L.k * L.k, s0, s1, c->eps, L.first ? 1 : 0,
                        L.wn, c->vel ? L.dwn : nullptr, c->stream, use_gauge ? L.a_in : nullptr, use_gauge ? L.beta : nullptr,
                        (use_gauge && c->prec != PREC_F16) ? L.b_sub : nullptr);
        if (use_gauge && c->prec == PREC_F16 && c->vel && L.b_sub && L.dwn_f)     // float16 model: the fused stages' version beside it
            launch_modulate(L.weight, L.sw, L.sb, L.cout, L.cin, L.k * L.k * L.k, s0, s1, c->eps, L.first ? 1 : 0,
                            L.wn, L.dwn_f, c->stream, L.a_in, nullptr, L.b_sub);
        launch_pack(L.wn, L.cout, L.cin, L.kind, L.pw, L.pw.w, c->stream);
        if (L.pwn.w) launch_pack(L.wn, L.cout, L.cin, L.kind, L.pwn, L.pwn.w, c->stream);
        if (L.pwn.dw) launch_pack(L.dwn, L.cout, L.cin, L.kind, L.pwn, L.pwn.dw, c->stream);
        if (c->vel && !(use_gauge && L.g6)) launch_pack(L.dwn, L.cout, L.cin, L.kind, L.pw, L.pw.dw, c->stream);
    }
    c->gauge_active = use_gauge;
    c->fuse = use_gauge && prec_is_half(c->prec);               // blocks with Layer::fskip run their skip inside conv_1
    if (pack_wino(c)) return 1;
    HIPCHK(hipGetLastError());
    c->modulated = true; c->mod_Om = Om; c->mod_Dz = Dz;
    ++c->epoch;                                                  // captured graphs hold the schedule of the previous modulation
    return 0;
}

int nbe_forward(nbe_ctx* c, const void* x, int D, int H, int W, float Dz, float vel_fac, void* disp, void* vel) {
    if (!c || !x || !disp) return fail("null argument");
    if (require_ready(c)) return 1;
    if (c->vel && !vel) return fail("velocity output pointer is NULL but compute_vel is set");
    if (check_dims(D, H, W)) return 1;
    HIPCHK(hipSetDevice(c->device));
    const int OD = D - 96, OH = H - 96, OW = W - 96;
    const int64_t in_bytes = (int64_t)c->in_chan * D * H * W * 4, out_bytes = (int64_t)c->out_chan * OD * OH * OW * 4;
    const bool xin_dev = is_device_ptr(x), out_dev = is_device_ptr(disp);
    c->slab = 0; c->pyx = false; c->pz = false;               // single inputs run on whole tensors, no periodicity
    if (ensure_workspace(c, D, H, W)) return 1;
    const float* xd = (const float*)x;
    if (!xin_dev) {
        if (in_bytes > c->box_in_bytes) { (void)hipFree(c->box_in); HIPCHK(hipMalloc((void**)&c->box_in, in_bytes)); c->box_in_bytes = in_bytes; }
        HIPCHK(hipMemcpyAsync(c->box_in, x, in_bytes, hipMemcpyHostToDevice, c->stream));
        xd = c->box_in;
    }
    char *dd = (char*)disp, *vd = (char*)vel;
    if (!out_dev) {
        const int64_t need = out_bytes * 2;
        if (need > c->box_out_bytes) { (void)hipFree(c->box_out); HIPCHK(hipMalloc((void**)&c->box_out, need)); c->box_out_bytes = need; }
        dd = c->box_out; vd = c->box_out + out_bytes;
    }
    if (prepare_range(c, xd, (int64_t)c->in_chan * D * H * W, Dz)) return 1;
    if (c->probe.on) {
        const int oe[3] = {OD, OH, OW};
        c->probe.tile = true;
        for (int d = 0; d < 3; ++d) {
            c->probe.o[d] = c->probe.p[d];
            if (c->probe.p[d] < 0 || c->probe.p[d] + c->probe.nout > oe[d]) return fail("branch probe: the block does not fit the output of this input");
        }
    }
    if (run_subbox(c, xd, D, H, W, 0, 0, 0, D, H, W, Dz, vel_fac, dd, vd, NBE_F32, OD, OH, OW, 0, 0, 0)) return 1;
    HIPCHK(hipGetLastError());
    if (!out_dev) {
        HIPCHK(hipMemcpyAsync(disp, dd, out_bytes, hipMemcpyDeviceToHost, c->stream));
        if (c->vel) HIPCHK(hipMemcpyAsync(vel, vd, out_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    if (!out_dev || !xin_dev) {
        HIPCHK(hipStreamSynchronize(c->stream));
        return check_range(c);                                  // host arrays: the call is synchronous anyway
    }
    return 0;
}

// Sub-boxes tiling the region [origin, origin + region) of a periodic box; results are written into an
// output array of spatial size `osize` at `oorigin` + the sub-box anchor inside the region.
int nbe_plan_tiles(const int64_t region[3], const int ndiv[3], int max_tile, int out_ndiv[3]) {
    if (!region || !ndiv || !out_ndiv) return fail("null argument");
    bool ok = max_tile > 0;
    for (int a = 0; a < 3 && ok; ++a) {
        if (ndiv[a] < 1 || region[a] < 1) return fail("sizes and ndiv must be positive");
        const int64_t crop = region[a] / ndiv[a];
        // merging is exact only when every anchor keeps the 2^3 stride lattice phase (crop % 8 == 0)
        // and nothing is left over (subbox.py:49 floors; the remainder stays zero)
        if (crop % 8 != 0 || crop * ndiv[a] != region[a]) ok = false;
    }
    for (int a = 0; a < 3; ++a) {
        out_ndiv[a] = ndiv[a];
        if (!ok) continue;
        const int64_t crop = region[a] / ndiv[a];
        int best = 1;
        for (int m = 1; m <= ndiv[a]; ++m)
            if (ndiv[a] % m == 0 && crop * m <= max_tile) best = m;
        out_ndiv[a] = ndiv[a] / best;
    }
    return 0;
}

// Schedule for a (D,H,W) input under a memory budget: 0 = whole tensors, S > 0 = z-slab schedule with S planes per
// slab (the deepest that fits), -1 = nothing fits.  *need receives the workspace bytes of the choice.
static int choose_slab(nbe_ctx* c, int D, int H, int W, int64_t budget, int64_t* need_out, bool pyx = false,
                       bool pz = false) {
    const int forced = c->slab_forced;
    const int keep = c->slab;
    const bool keep_p = c->pyx, keep_z = c->pz;
    int result = -1;
    c->pyx = pyx; c->pz = pyx && pz;
    // deeper slabs than 128 planes buy < 1 % (the 2-plane overlaps are already < 5 % there) for tens of GB of workspace
    const int cand[4] = {0, 128, 64, 32};
    for (int i = 0; i < 4 && result < 0; ++i) {
        int S = cand[i];
        if (pyx && S == 0) { if (forced == 0) break; continue; }  // periodic-yx exists only in the slab schedule
        if (forced == 0 && S != 0) break;
        if (forced > 0) { if (i > (pyx ? 1 : 0)) break; S = forced & ~1; }
        if (!pyx && S > 0 && D - 8 <= S) continue;                // a single slab is the whole-tensor schedule
        c->slab = S;
        const int64_t need = workspace_need(c, D, H, W);
        if (need >= 0 && need <= budget) { result = S; if (need_out) *need_out = need; }
    }
    c->slab = keep;
    c->pyx = keep_p; c->pz = keep_z;
    return result;
}

// The grid process_region will run: among all merges of the caller's sub-boxes (exact only when crop % 8 == 0 on
// every axis) with tile edge <= max_tile, the one with the largest tile volume whose workspace fits the device
// memory that is free now (plus what this context already holds, minus `reserve`); ties go to the tile that is
// longest along the last (fastest) axis.  512^3 / ndiv 4 on a 288 GB MI355X: four tiles of 256 x 256 x 512
// (input 352 x 352 x 608, a ~175 GB workspace), 12 % fewer FLOPs than eight tiles of 256^3.
static int64_t plan_budget(nbe_ctx* c, int64_t reserve) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return -1; }
    // NBE_MEM_FRACTION (default 1): share of the free memory this context may plan with -- for rigs that run several
    // ranks on one card, where every rank sees the same free memory at the same time
    static const double frac = getenv("NBE_MEM_FRACTION") ? std::min(1.0, std::max(0.01, atof(getenv("NBE_MEM_FRACTION")))) : 1.0;
    return (int64_t)(((double)free_b + (double)c->ws_bytes) * frac) - reserve - ((int64_t)3 << 30);
}

static int plan_tiles_mem(nbe_ctx* c, const int64_t region[3], const int ndiv[3], int64_t reserve, bool full_yx,
                          bool full_z, int out_ndiv[3]) {
    for (int a = 0; a < 3; ++a) out_ndiv[a] = ndiv[a];
    c->slab = 0;
    c->pyx = false; c->pz = false;
    if (!c->have_weights) return 0;
    full_yx = full_yx && c->pyx_allowed;
    const int64_t budget = plan_budget(c, reserve);
    if (budget < 0) return 0;
    // schedule of a tile of e0 x e1 x e2 output voxels: periodic-yx when it spans the box in y and x, else padded
    auto schedule = [&](int64_t e0, int64_t e1, int64_t e2, bool spans, int* slab, bool* pyx) -> bool {
        if (spans && full_yx && !check_dims_pyx((int)e0 + 96, (int)e1 + 2, (int)e2 + 2)) {
            const int sl = choose_slab(c, (int)e0 + 96, (int)e1 + 2, (int)e2 + 2, budget, nullptr, true, full_z && e0 == region[0]);
            if (sl > 0) { *slab = sl; *pyx = true; return true; }
        }
        if (check_dims((int)e0 + 96, (int)e1 + 96, (int)e2 + 96)) return false;
        const int sl = choose_slab(c, (int)e0 + 96, (int)e1 + 96, (int)e2 + 96, budget, nullptr, false);
        if (sl < 0) return false;
        *slab = sl; *pyx = false;
        return true;
    };
    int best_slab = 0;
    bool best_pyx = false;
    schedule(region[0] / ndiv[0], region[1] / ndiv[1], region[2] / ndiv[2], ndiv[1] == 1 && ndiv[2] == 1 &&
             region[1] % ndiv[1] == 0 && region[2] % ndiv[2] == 0, &best_slab, &best_pyx);       // the caller's own grid
    (void)nbe_last_error();
    c->slab = best_slab; c->pyx = best_pyx;
    c->pz = best_pyx && full_z && ndiv[0] == 1;
    if (c->max_tile <= 0) return 0;
    int64_t crop[3];
    for (int a = 0; a < 3; ++a) {
        if (ndiv[a] < 1 || region[a] < 1) return fail("sizes and ndiv must be positive");
        crop[a] = region[a] / ndiv[a];
        if (crop[a] % 8 != 0 || crop[a] * ndiv[a] != region[a]) return 0;      // merging would not be exact
    }
    int64_t best_vol = 0, best_w = 0, miss_vol = 0, miss_need = 0;
    for (int m0 = 1; m0 <= ndiv[0]; ++m0) {
        if (ndiv[0] % m0 || crop[0] * m0 > c->max_tile) continue;
        for (int m1 = 1; m1 <= ndiv[1]; ++m1) {
            if (ndiv[1] % m1 || crop[1] * m1 > c->max_tile) continue;
            for (int m2 = 1; m2 <= ndiv[2]; ++m2) {
                if (ndiv[2] % m2 || crop[2] * m2 > c->max_tile) continue;
                const int64_t e0 = crop[0] * m0, e1 = crop[1] * m1, e2 = crop[2] * m2, vol = e0 * e1 * e2;
                const int64_t w = e2 * 1000000 + e1 * 1000 + e0;               // tie-break: long last axis
                if (vol < best_vol || (vol == best_vol && w <= best_w)) continue;
                int sl = 0; bool px = false;
                if (!schedule(e0, e1, e2, m1 == ndiv[1] && m2 == ndiv[2], &sl, &px)) {
                    (void)nbe_last_error();
                    if (vol > miss_vol) {                        // what the largest merge that did not fit would have needed
                        const bool spans = m1 == ndiv[1] && m2 == ndiv[2] && full_yx;
                        const bool kp = c->pyx, kz = c->pz; const int ks_ = c->slab;
                        c->pyx = spans; c->pz = spans && full_z && e0 == region[0]; c->slab = 32;
                        const int ext = spans ? 2 : 96;
                        const int64_t need = workspace_need(c, (int)e0 + 96, (int)e1 + ext, (int)e2 + ext);
                        c->pyx = kp; c->pz = kz; c->slab = ks_;
                        (void)nbe_last_error();
                        if (need > 0) { miss_vol = vol; miss_need = need; }
                    }
                    continue;
                }
                best_vol = vol; best_w = w; best_slab = sl; best_pyx = px;
                out_ndiv[0] = ndiv[0] / m0; out_ndiv[1] = ndiv[1] / m1; out_ndiv[2] = ndiv[2] / m2;
            }
        }
    }
    c->slab = best_slab; c->pyx = best_pyx;
    c->pz = best_pyx && full_z && out_ndiv[0] == 1;
    c->plan_tiles = out_ndiv[0] * out_ndiv[1] * out_ndiv[2];
    c->plan_short_gb = 0.0;
    if (miss_vol > best_vol) {                                   // a larger exact merge exists and only memory kept the planner from it
        c->plan_short_gb = std::max(0.0, (double)(miss_need - budget) / 1e9);
        static const bool quiet = getenv("NBE_QUIET") && atoi(getenv("NBE_QUIET")) != 0;
        const int64_t key = miss_vol * 1000 + c->plan_tiles;
        if (!quiet && key != c->plan_logged) {
            fprintf(stderr, "nbe: box %lld x %lld x %lld runs as %d x %d x %d tiles: a larger tile needs at least %.0f GB of workspace (even in "
                            "32-plane slabs), %.0f GB are free for it on device %d -- expect 1.1 - 1.4 x the time of the larger plan\n",
                    (long long)region[0], (long long)region[1], (long long)region[2], out_ndiv[0], out_ndiv[1], out_ndiv[2],
                    miss_need / 1e9, budget / 1e9, c->device);
            c->plan_logged = key;
        }
    }
    return 0;
}

int nbe_plan_tiles_ctx(nbe_ctx* c, const int64_t region[3], const int ndiv[3], int periodic_box, int out_ndiv[3]) {
    if (!c || !region || !ndiv || !out_ndiv) return fail("null argument");
    HIPCHK(hipSetDevice(c->device));
    return plan_tiles_mem(c, region, ndiv, 0, periodic_box != 0, periodic_box != 0, out_ndiv);
}

int nbe_set_precision(nbe_ctx* c, int prec) {
    if (!c) return fail("null context");
    if (prec != PREC_F32 && prec != PREC_F16X3 && prec != PREC_F16) return fail("precision must be NBE_PREC_F32 (0), NBE_PREC_F16X3 (1) or NBE_PREC_F16 (2)");
    if (c->have_weights && prec != c->prec)
        return fail("nbe_set_precision must be called before the weights are loaded (they are packed per precision)");
    c->prec = prec;
    return 0;
}

int nbe_set_periodic(nbe_ctx* c, int on) {
    if (!c) return fail("null context");
    c->pyx_allowed = on != 0;
    return 0;
}

int nbe_set_slab(nbe_ctx* c, int slab) {
    if (!c) return fail("null context");
    if (slab > 0 && (slab & 1)) return fail("slab must be even");
    c->slab_forced = slab < 0 ? -1 : slab;
    return 0;
}

int nbe_set_max_tile(nbe_ctx* c, int max_tile) {
    if (!c) return fail("null context");
    if (max_tile < 0) return fail("max_tile must be >= 0 (0 = keep the caller's sub-box grid)");
    c->max_tile = max_tile;
    return 0;
}

static int process_region(nbe_ctx* c, const void* box, const int64_t bsize[3], const int64_t origin[3],
                          const int64_t region[3], const int ndiv_in[3], const int* order, int norder,
                          float Dz, float vel_fac, void* disp, void* vel, int out_dtype,
                          const int64_t osize[3], const int64_t oorigin[3], bool zero_out,
                          nbe_progress_cb cb, void* user) {
    if (require_ready(c)) return 1;
    if (c->vel && !vel) return fail("velocity output pointer is NULL but compute_vel is set");
    if (out_dtype != NBE_F32 && out_dtype != NBE_F16) return fail("out_dtype must be NBE_F32 or NBE_F16");
    for (int i = 0; i < 3; ++i) {
        if (ndiv_in[i] < 1 || bsize[i] < 1 || region[i] < 1 || osize[i] < 1) return fail("sizes and ndiv must be positive");
        if (bsize[i] > 2000000000LL / 4 || osize[i] > 2000000000LL / 4) return fail("box axis too large");
    }
    const int S0 = (int)bsize[0], S1 = (int)bsize[1], S2 = (int)bsize[2];
    const int O0 = (int)osize[0], O1 = (int)osize[1], O2 = (int)osize[2];
    // Internal tiling: with an explicit sub-box list the caller's grid is used as given; otherwise adjacent
    // sub-boxes may be merged into larger tiles (nbe_plan_tiles) -- identical results, less halo recompute.
    int ndiv_eff[3] = {ndiv_in[0], ndiv_in[1], ndiv_in[2]};
    {
        HIPCHK(hipSetDevice(c->device));
        // host arrays in / out are staged in device buffers that are allocated below: keep room for them
        const int64_t in_b = (int64_t)S0 * S1 * S2 * c->in_chan * 4;
        const int64_t out_b = (int64_t)O0 * O1 * O2 * c->out_chan * (out_dtype == NBE_F16 ? 2 : 4) * (c->vel ? 2 : 1);
        const int64_t reserve = (is_device_ptr(box) ? 0 : std::max<int64_t>(0, in_b - c->box_in_bytes)) +
                                (is_device_ptr(disp) ? 0 : std::max<int64_t>(0, out_b - c->box_out_bytes));
        // the region is the periodic box itself in y and x: tiles that span it may run in periodic-yx mode
        const bool full_yx = c->pyx_allowed && origin[1] == 0 && origin[2] == 0 && region[1] == bsize[1] && region[2] == bsize[2];
        const bool full_z = origin[0] == 0 && region[0] == bsize[0];        // ... and in z
        auto tile_dims = [&](int* d, int* h, int* w) {            // input dims of a tile of the current grid / mode
            const int ext = c->pyx ? 2 : 96;
            *d = (int)(region[0] / ndiv_eff[0]) + 96; *h = (int)(region[1] / ndiv_eff[1]) + ext; *w = (int)(region[2] / ndiv_eff[2]) + ext;
        };
        // schedule (whole tensors or z-slabs, padded or periodic-yx) of a given grid under the memory that is free now
        auto schedule_for_grid = [&]() {
            c->slab = 0; c->pyx = false; c->pz = false;
            const int64_t budget = plan_budget(c, reserve);
            if (budget < 0 || !c->have_weights) return;
            const int e0 = (int)(region[0] / ndiv_eff[0]), e1 = (int)(region[1] / ndiv_eff[1]), e2 = (int)(region[2] / ndiv_eff[2]);
            if (full_yx && ndiv_eff[1] == 1 && ndiv_eff[2] == 1 && !check_dims_pyx(e0 + 96, e1 + 2, e2 + 2)) {
                const bool z1 = full_z && ndiv_eff[0] == 1;
                const int sl = choose_slab(c, e0 + 96, e1 + 2, e2 + 2, budget, nullptr, true, z1);
                if (sl > 0) { c->slab = sl; c->pyx = true; c->pz = z1; return; }
            }
            (void)nbe_last_error();
            const int d = e0 + 96, h = e1 + 96, w = e2 + 96;
            if (d >= 104 && h >= 104 && w >= 104 && !(d % 8) && !(h % 8) && !(w % 8)) {
                const int sl = choose_slab(c, d, h, w, budget, nullptr);
                if (sl > 0) c->slab = sl;
                else if (sl < 0 && c->slab_forced < 0 && d - 8 > 32) c->slab = 32;   // nothing fits the budget: smallest footprint
            }
        };
        if (order) schedule_for_grid();                          // explicit sub-box list: the caller's grid as given
        else if (plan_tiles_mem(c, region, ndiv_in, reserve, full_yx, full_z, ndiv_eff)) return 1;
        // fall back to cubic tiles <= 256, then to the caller's grid, when the workspace cannot be allocated after all
        for (int attempt = 0; attempt < 2 && !order; ++attempt) {
            if (ndiv_eff[0] == ndiv_in[0] && ndiv_eff[1] == ndiv_in[1] && ndiv_eff[2] == ndiv_in[2]) break;
            int d, h, w; tile_dims(&d, &h, &w);
            if (!ensure_workspace(c, d, h, w)) break;
            (void)hipGetLastError();
            if (attempt == 0 && c->max_tile > 256) { if (nbe_plan_tiles(region, ndiv_in, 256, ndiv_eff)) return 1; }
            else { ndiv_eff[0] = ndiv_in[0]; ndiv_eff[1] = ndiv_in[1]; ndiv_eff[2] = ndiv_in[2]; }
            schedule_for_grid();
        }
    }
    const int* ndiv = ndiv_eff;
    const int c0 = (int)(region[0] / ndiv[0]), c1 = (int)(region[1] / ndiv[1]), c2 = (int)(region[2] / ndiv[2]);   // subbox.py:49 (floor)
    const int hal = c->pyx ? 1 : 48;                              // y/x context gathered with the tile
    const int D = c0 + 96, H = c1 + 2 * hal, W = c2 + 2 * hal;
    if (c->pyx ? check_dims_pyx(D, H, W) : check_dims(D, H, W)) return 1;
    for (int i = 0; i < 3; ++i)
        if (oorigin[i] < 0 || oorigin[i] + region[i] > osize[i]) return fail("output region does not fit the output array");
    HIPCHK(hipSetDevice(c->device));
    const int64_t in_bytes = (int64_t)S0 * S1 * S2 * c->in_chan * 4;
    const int esz = out_dtype == NBE_F16 ? 2 : 4;
    const int64_t out_bytes = (int64_t)O0 * O1 * O2 * c->out_chan * esz;
    const bool in_dev = is_device_ptr(box), out_dev = is_device_ptr(disp);
    if (ensure_workspace(c, D, H, W)) return 1;
    const float* bd = (const float*)box;
    // Host arrays in and out, the whole periodic box as one tile in the z-slab schedule, pinned outputs: pipelined
    // (HostPipe).  Anything else: the box goes up in one piece before the first tile and the fields come down after the last.
    const bool pipe_off = getenv("NBE_HOST_PIPE") && atoi(getenv("NBE_HOST_PIPE")) == 0;   // read per call: A/B in one process
    auto& P = c->pipe;
    P.active = false;
    P.tiles = false; P.slabwise = false;
    const bool whole_box = S0 == O0 && S1 == O1 && S2 == O2 && oorigin[0] == 0 && oorigin[1] == 0 && oorigin[2] == 0 &&
                           origin[0] == 0 && origin[1] == 0 && origin[2] == 0 && region[0] == S0 && region[1] == S1 && region[2] == S2;
    const bool one_tile = c->slab > 0 && c->pyx && c->pz && ndiv[0] * ndiv[1] * ndiv[2] == 1;
    // several tiles that cover the box exactly (nothing left for the zeros of subbox.py:168-170): pipelined tile by tile
    const bool many = !one_tile && (int64_t)c0 * ndiv[0] == S0 && (int64_t)c1 * ndiv[1] == S1 && (int64_t)c2 * ndiv[2] == S2;
    if (!in_dev && !out_dev && !order && !pipe_off && whole_box && (one_tile || many)) {
        P.active = one_tile; P.tiles = many;
        P.hbox = (const float*)box; P.in_pinned = is_pinned_host_ptr(box);
        P.C = c->in_chan; P.S0 = S0; P.S1 = S1; P.S2 = S2; P.o0 = (int)origin[0] - 48;
        P.up.assign(S0, 0); P.gz = 0; P.nstage = 0;
        P.out_async = is_pinned_host_ptr(disp) && (!c->vel || is_pinned_host_ptr(vel));
        P.hdisp = (char*)disp; P.hvel = c->vel ? (char*)vel : nullptr; P.esz = esz; P.O0 = O0; P.O1 = O1; P.O2 = O2;
        if (!c->up_stream) {
            HIPCHK(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
            HIPCHK(hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_down, hipEventDisableTiming));
            for (int i = 0; i < nbe_ctx::NSTAGE; ++i) HIPCHK(hipEventCreateWithFlags(&c->stage_free[i], hipEventDisableTiming));
        }
        const int64_t sb = (int64_t)PIPE_CHUNK * S1 * S2 * 4 * c->in_chan;
        if (!P.in_pinned && sb > c->stage_bytes) {
            for (int i = 0; i < nbe_ctx::NSTAGE; ++i) {
                if (c->stage_buf[i]) (void)hipHostFree(c->stage_buf[i]);
                c->stage_buf[i] = nullptr;
                HIPCHK(hipHostMalloc((void**)&c->stage_buf[i], sb, hipHostMallocDefault));
            }
            c->stage_bytes = sb;
        }
    }
    struct PipeGuard { nbe_ctx::HostPipe& p; ~PipeGuard() { p.active = false; p.tiles = false; p.slabwise = false; } } pipe_guard{P};
    c->last_piped = P.active || P.tiles;
    const bool piped = P.active || P.tiles;
    if (!in_dev) {
        if (in_bytes > c->box_in_bytes) { (void)hipFree(c->box_in); c->box_in = nullptr; c->box_in_bytes = 0;
                                          HIPCHK(hipMalloc((void**)&c->box_in, in_bytes)); c->box_in_bytes = in_bytes; }
        if (piped) {
            // whatever still reads the device box from the previous call must have finished before the uploads start
            HIPCHK(hipStreamSynchronize(c->stream));
        } else {
            HIPCHK(hipMemcpyAsync(c->box_in, box, in_bytes, hipMemcpyHostToDevice, c->stream));
        }
        bd = c->box_in;
    }
    char *dd = (char*)disp, *vd = (char*)vel;
    if (!out_dev) {
        const int64_t need = out_bytes * (c->vel ? 2 : 1);
        if (need > c->box_out_bytes) { (void)hipFree(c->box_out); c->box_out = nullptr; c->box_out_bytes = 0;
                                       HIPCHK(hipMalloc((void**)&c->box_out, need)); c->box_out_bytes = need; }
        dd = c->box_out; vd = c->box_out + out_bytes;
        P.ddisp = dd; P.dvel = c->vel ? vd : nullptr;
    }
    // subbox.py:168-170: outputs start as zeros (voxels beyond ndiv*crop_size stay zero); the one-tile plan of the
    // pipelined path writes every voxel
    if ((zero_out || !out_dev) && !piped) {
        HIPCHK(hipMemsetAsync(dd, 0, out_bytes, c->stream));
        if (c->vel) HIPCHK(hipMemsetAsync(vd, 0, out_bytes, c->stream));
    }
    const bool trace = piped && getenv("NBE_PIPE_TRACE") && atoi(getenv("NBE_PIPE_TRACE")) != 0;
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_up0 = 0, t_range = 0, t_enq = 0;
    std::unique_ptr<Progress> reporter;
    if (cb) reporter.reset(new Progress(cb, user, c->device));
    struct ProgGuard { nbe_ctx* c; ~ProgGuard() { c->prog = nullptr; c->prog_cb = nullptr; } } prog_guard{c};
    c->prog = reporter.get();
    // tiles in the z-slab schedule gather slab by slab as their planes land (the first tile of a 1024^3 box would otherwise
    // wait for 608 planes = 7.6 GB before its first kernel)
    P.slabwise = P.tiles && c->slab > 0;
    if (P.tiles) {
        // the first tile's (first slab's) planes, then max|x| on the host while the DMA runs
        P.o0 = -48;
        const int zs = c->pyx && c->pz ? 40 : 0;
        if (P.slabwise ? pipe_upload(c, zs, zs + std::min(c->slab, PIPE_EDGE) + 8) : pipe_upload(c, 0, D)) return 1;
        t_up0 = ms_since();
        if (prepare_range(c, nullptr, (int64_t)c->in_chan * S0 * S1 * S2, Dz, (const float*)box)) return 1;
        t_range = ms_since();
    } else if (P.active) {
        // start the first upload, then reduce max|x| on the host while the DMA runs
        if (pipe_upload(c, 40, 40 + std::min(c->slab, PIPE_EDGE) + 8)) return 1;
        t_up0 = ms_since();
        if (prepare_range(c, nullptr, (int64_t)c->in_chan * S0 * S1 * S2, Dz, (const float*)box)) return 1;
        t_range = ms_since();
    } else if (prepare_range(c, bd, (int64_t)c->in_chan * S0 * S1 * S2, Dz)) return 1;
    const int total = ndiv[0] * ndiv[1] * ndiv[2];
    const int n = order ? norder : total;
    for (int k = 0; k < n; ++k) {
        const int idx = order ? order[k] : k;
        if (idx < 0 || idx >= total) return fail("sub-box index %d out of range (0..%d)", idx, total - 1);
        // subbox.py:60-66: row-major over ndiv, last axis fastest
        const int a0 = (idx / (ndiv[1] * ndiv[2])) * c0, a1 = ((idx / ndiv[2]) % ndiv[1]) * c1, a2 = (idx % ndiv[2]) * c2;
        c->prog_cb = cb; c->prog_user = user; c->prog_k = k; c->prog_n = n;
        if (P.tiles) {                                           // this tile's planes (most were sent under the tile before)
            P.o0 = a0 - 48; P.o1 = a1 - hal; P.o2 = a2 - hal; P.gz = 0;
            if (!P.slabwise) {
                if (pipe_upload(c, 0, D)) return 1;
                HIPCHK(hipEventRecord(c->ev_up, c->up_stream));
                HIPCHK(hipStreamWaitEvent(c->stream, c->ev_up, 0));
            }
        }
        if (c->probe.on) {                                       // branch probe: armed for the tile that holds the block
            auto& Pb = c->probe;
            const int ta[3] = {(int)oorigin[0] + a0, (int)oorigin[1] + a1, (int)oorigin[2] + a2}, te[3] = {c0, c1, c2};
            Pb.tile = true;
            for (int d = 0; d < 3; ++d) {
                Pb.o[d] = Pb.p[d] - ta[d];
                if (Pb.o[d] < 0 || Pb.o[d] + Pb.nout > te[d] || Pb.o[d] % 8 != 0) Pb.tile = false;
            }
        }
        if (run_tile(c, bd, S0, S1, S2, (int)origin[0] + a0 - 48, (int)origin[1] + a1 - hal, (int)origin[2] + a2 - hal,
                       D, H, W, Dz, vel_fac, dd, vd, out_dtype, O0, O1, O2,
                       (int)oorigin[0] + a0, (int)oorigin[1] + a1, (int)oorigin[2] + a2)) return 1;
        if (P.tiles) {
            if (P.out_async && pipe_output(c, a0, c0, a1, a2, c1, c2)) return 1;
            if (k + 1 < n) {                                     // the next tile's planes go up under this tile's kernels
                const int idn = k + 1;
                P.o0 = (idn / (ndiv[1] * ndiv[2])) * c0 - 48;
                if (pipe_upload(c, 0, D)) return 1;              // (what this tile's slabs have not already brought up)
            }
        }
        if (reporter) reporter->post(piped && P.out_async ? c->down_stream : c->stream, (k + 1) * 1000, n * 1000);
    }
    c->prog_cb = nullptr;
    HIPCHK(hipGetLastError());
    if (!out_dev && !(piped && P.out_async)) {
        HIPCHK(hipMemcpyAsync(disp, dd, out_bytes, hipMemcpyDeviceToHost, c->stream));
        if (c->vel) HIPCHK(hipMemcpyAsync(vel, vd, out_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    if (!out_dev || !in_dev) {
        t_enq = ms_since();
        HIPCHK(hipStreamSynchronize(c->stream));
        const double t_comp = ms_since();
        if (piped) { HIPCHK(hipStreamSynchronize(c->up_stream)); HIPCHK(hipStreamSynchronize(c->down_stream)); }
        reporter.reset();                                        // every report has been delivered when this returns
        if (trace)
            fprintf(stderr, "nbe pipe: first upload staged %.1f ms, max|x| on %d host threads %.1f ms, all work enqueued %.1f ms, "
                            "kernels done %.1f ms, last slab on the host %.1f ms (input %s, outputs %s)\n",
                    t_up0, host_threads(), t_range - t_up0, t_enq, t_comp, ms_since(), P.in_pinned ? "pinned" : "pageable",
                    P.out_async ? "pinned" : "pageable");
        return check_range(c);                                  // host arrays: the call is synchronous anyway
    }
    if (reporter) { HIPCHK(hipStreamSynchronize(c->stream)); reporter.reset(); }   // a call with a progress callback is synchronous
    return 0;
}

int nbe_process_box(nbe_ctx* c, const void* box, const int64_t size[3], const int ndiv[3], const int pad[6],
                    float Dz, float vel_fac, void* disp, void* vel, int out_dtype, nbe_progress_cb cb, void* user) {
    if (!c || !box || !disp || !size || !ndiv || !pad) return fail("null argument");
    for (int i = 0; i < 6; ++i)
        if (pad[i] != 48) return fail("padding must be 48 on every side (receptive field of the network, subbox.py:43); got %d", pad[i]);
    const int64_t zero[3] = {0, 0, 0};
    return process_region(c, box, size, zero, size, ndiv, nullptr, 0, Dz, vel_fac, disp, vel, out_dtype, size, zero,
                          true, cb, user);
}

int nbe_process_region(nbe_ctx* c, const void* box, const int64_t box_size[3], const int64_t origin[3],
                       const int64_t region[3], const int ndiv[3], const int* order, int norder,
                       float Dz, float vel_fac, void* disp, void* vel, int out_dtype,
                       const int64_t out_size[3], const int64_t out_origin[3]) {
    if (!c || !box || !disp || !box_size || !origin || !region || !ndiv || !out_size || !out_origin) return fail("null argument");
    return process_region(c, box, box_size, origin, region, ndiv, order, norder, Dz, vel_fac, disp, vel, out_dtype,
                          out_size, out_origin, false, nullptr, nullptr);
}

void* nbe_host_alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_free.lower_bound(bytes);
        if (it != g_pin_free.end() && it->first <= bytes + bytes / 8) {      // close enough in size: reuse
            void* p = it->second; const size_t sz = it->first;
            g_pin_free.erase(it); g_pin_free_bytes -= sz; g_pin_live[p] = sz;
            return p;
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fail("nbe_host_alloc: hipHostMalloc(%zu) failed", bytes); return nullptr; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_live[p] = bytes;
    return p;
}

int nbe_host_free(void* p) {
    if (!p) return 0;
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_live.find(p);
        if (it == g_pin_live.end()) return fail("nbe_host_free: %p was not allocated by nbe_host_alloc", p);
        sz = it->second; g_pin_live.erase(it);
        if (g_pin_free_bytes + sz <= pin_pool_cap()) { g_pin_free.emplace(sz, p); g_pin_free_bytes += sz; return 0; }
    }
    (void)hipHostFree(p);
    return 0;
}

int nbe_host_trim(void) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto& kv : g_pin_free) (void)hipHostFree(kv.second);
    g_pin_free.clear(); g_pin_free_bytes = 0;
    return 0;
}

// ---- brick mode: one rank's z-slab of a periodic box, the context below the full-resolution level exchanged ---------
static constexpr int BRICK_RAW = 4;      // planes of RAW input a brick needs from either z neighbour (the level-0 encoder's reach for the brick's own planes)
static int brick_setup(nbe_ctx* c, const int64_t bsize[3], int* D, int* H, int* W, int64_t* need_out = nullptr) {
    c->sst.valid = false;
    if (require_ready(c)) return 1;
    if (!bsize) return fail("null argument");
    const int64_t b0 = bsize[0], S1 = bsize[1], S2 = bsize[2];
    if (b0 % 8 != 0 || b0 < 48) return fail("brick depth %lld unsupported: a multiple of 8, at least 48", (long long)b0);
    *D = (int)b0 + 96; *H = (int)S1 + 2; *W = (int)S2 + 2;
    if (check_dims_pyx(*D, *H, *W)) return 1;
    HIPCHK(hipSetDevice(c->device));
    c->pyx = true; c->pz = false; c->zx = true;
    // a brick of this size has been planned before and its workspace is still held (nbe_brick_plan allocates it): the same plan,
    // whatever the other tenants of the card have allocated since
    if (c->bp_slab > 0 && c->bp_size[0] == b0 && c->bp_size[1] == S1 && c->bp_size[2] == S2 && c->ws_bytes >= c->bp_need) {
        c->slab = c->bp_slab;
        if (need_out) *need_out = c->bp_need;
        return 0;
    }
    const int64_t budget = plan_budget(c, 0);
    int64_t need = 0;
    const int sl = choose_slab(c, *D, *H, *W, budget < 0 ? INT64_MAX / 4 : budget, &need, true, false);
    if (sl <= 0) { c->zx = false; return fail("brick of %lld x %lld x %lld does not fit the device memory that is free", (long long)b0, (long long)S1, (long long)S2); }
    c->slab = sl; c->pyx = true; c->pz = false;
    c->bp_size[0] = b0; c->bp_size[1] = S1; c->bp_size[2] = S2; c->bp_slab = sl; c->bp_need = need;
    if (need_out) *need_out = need;
    return 0;
}

int64_t nbe_brick_halo_bytes(nbe_ctx* c, const int64_t bsize[3], int which) {
    if (!c || !bsize || which < 0 || which > 3) return -1;
    if (which == 3) return brick_halo_bytes(c, BRICK_H0, (int)bsize[1] + 2, (int)bsize[2] + 2);   // whole planes, wrap-around columns included
    if (which == 0) return (int64_t)c->in_chan * BRICK_RAW * bsize[1] * bsize[2] * 4;       // raw input planes, float32
    if (which == 1) return brick_halo_bytes(c, BRICK_H1, (int)bsize[1] / 2, (int)bsize[2] / 2) + 16;   // + the sender's range shift
    return brick_halo_bytes(c, BRICK_H2, (int)bsize[1] / 4, (int)bsize[2] / 4);
}

// > 0: the brick fits the device memory that is free now, with that many planes per z-slab; 0: it does not (no error)
int nbe_brick_plan(nbe_ctx* c, const int64_t bsize[3]) {
    if (!c || !bsize) return 0;
    int D, H, W;
    const int keep_slab = c->slab; const bool kp = c->pyx, kz = c->pz;
    int rc = brick_setup(c, bsize, &D, &H, &W);
    // take the workspace now: what is free when the first brick is encoded may be less (other ranks of a shared card, the
    // caller's exchange buffers), and the ranks must not part ways after they have agreed on brick mode
    if (!rc) { rc = ensure_workspace(c, D, H, W); if (rc) { c->bp_slab = 0; (void)hipGetLastError(); } }
    const int sl = rc ? 0 : c->slab;
    c->zx = false; c->slab = keep_slab; c->pyx = kp; c->pz = kz;
    (void)nbe_last_error();
    return sl;
}

struct BrickOff { nbe_ctx* c; ~BrickOff() { c->phase = 0; c->zx = false; c->bio.skip_ready = nullptr; c->bio.skip_recv_lo = nullptr; } };

int nbe_brick_encode(nbe_ctx* c, const void* box, const int64_t bsize[3], float Dz, float vel_fac, void* send_lo, void* send_hi,
                     void* skip_send_lo, void* skip_send_hi) {
    if (!c || !box || !send_lo || !send_hi || !skip_send_lo || !skip_send_hi) return fail("null argument");
    if (!is_device_ptr(box) || !is_device_ptr(send_lo) || !is_device_ptr(send_hi) || !is_device_ptr(skip_send_lo) || !is_device_ptr(skip_send_hi))
        return fail("nbe_brick_encode takes device pointers");
    int D, H, W;
    if (brick_setup(c, bsize, &D, &H, &W)) return 1;
    BrickOff off{c};
    if (ensure_workspace(c, D, H, W)) return 1;
    const int Dh = (int)bsize[0] + 2 * BRICK_RAW;
    if (prepare_range(c, (const float*)box, (int64_t)c->in_chan * Dh * bsize[1] * bsize[2], Dz)) return 1;
    c->arena.reset();
    Tensor tin = talloc(c, c->in_chan, D, H, W);
    tin.pad = 1; set_org(tin, 0, 48, 48);
    // the haloed brick is (C, b0 + 8, S1, S2): planes [44, b0 + 52) of the tile's frame -- all the level-0 encoder reads for the
    // brick's own planes of the skip connection (the head reads the brick's own input planes); y and x periodic (origin -1)
    launch_gather((const float*)box, c->in_chan, Dh, (int)bsize[1], (int)bsize[2], 0, -1, -1, zview(tin, 48 - BRICK_RAW, Dh).p,
                  Dz / 6.0f * c->act_scale, c->prec, c->stream);
    c->phase = 1; c->bio.send_lo = send_lo; c->bio.send_hi = send_hi; c->bio.skip_send_lo = skip_send_lo; c->bio.skip_send_hi = skip_send_hi;
    c->sst.D = D; c->sst.H = H; c->sst.W = W;
    c->sst.Dz = Dz; c->sst.vel_fac = vel_fac; c->sst.act_scale = c->act_scale; c->sst.ws = c->ws;
    const HeadOut ho{nullptr, nullptr, NBE_F32, (int)bsize[0], (int)bsize[1], (int)bsize[2], 0, 0, 0, Dz, vel_fac};
    if (network_stream(c, tin, ho, c->slab)) return 1;
    // the last word of either face carries this rank's range shift: the receiver refuses faces computed with another one
    const int64_t body = brick_halo_bytes(c, BRICK_H1, (int)bsize[1] / 2, (int)bsize[2] / 2);
    unsigned bits; memcpy(&bits, &c->act_scale, 4);
    launch_tag_word((unsigned*)((char*)send_lo + body), bits, nullptr, nullptr, 0, c->stream);
    launch_tag_word((unsigned*)((char*)send_hi + body), bits, nullptr, nullptr, 0, c->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

static int brick_resume(nbe_ctx* c, int phase) {
    if (!c->sst.valid) return fail("brick call without a preceding nbe_brick_encode (or another call has used this context's workspace in between)");
    if (c->sst.ws != c->ws) return fail("the workspace was reallocated since nbe_brick_encode");
    HIPCHK(hipSetDevice(c->device));
    c->pyx = true; c->pz = false; c->zx = true; c->phase = phase;
    return 0;
}

int nbe_brick_interior(nbe_ctx* c) {
    if (!c) return fail("null context");
    if (brick_resume(c, 2)) return 1;
    BrickOff off{c};
    const HeadOut ho{};
    if (network_stream(c, c->sst.tin, ho, c->sst.S)) return 1;
    HIPCHK(hipGetLastError());
    return 0;
}

int nbe_brick_exchange(nbe_ctx* c, const void* recv_lo, const void* recv_hi, void* send2_lo, void* send2_hi) {
    if (!c || !recv_lo || !recv_hi || !send2_lo || !send2_hi) return fail("null argument");
    if (brick_resume(c, 3)) return 1;
    BrickOff off{c};
    c->bio.recv_lo = recv_lo; c->bio.recv_hi = recv_hi; c->bio.send_lo = send2_lo; c->bio.send_hi = send2_hi;
    if (c->flags) {                                               // (strict float32 contexts have no range shift: nothing to compare)
        const int64_t body = brick_halo_bytes(c, BRICK_H1, (c->sst.H - 2) / 2, (c->sst.W - 2) / 2);
        unsigned bits; memcpy(&bits, &c->sst.act_scale, 4);
        launch_tag_word(nullptr, bits, (const unsigned*)((const char*)recv_lo + body), c->flags + 1, 2u, c->stream);
        launch_tag_word(nullptr, bits, (const unsigned*)((const char*)recv_hi + body), c->flags + 1, 2u, c->stream);
    }
    const HeadOut ho{};
    if (network_stream(c, c->sst.tin, ho, c->sst.S)) return 1;
    HIPCHK(hipGetLastError());
    return 0;
}

int nbe_brick_finish(nbe_ctx* c, const void* recv_lo, const void* recv_hi, const void* skip_recv_lo, const void* skip_recv_hi,
                     void* skip_ready_event, float Dz, float vel_fac, void* disp, void* vel, int out_dtype) {
    if (!c || !recv_lo || !recv_hi || !skip_recv_lo || !skip_recv_hi || !disp) return fail("null argument");
    if (c->vel && !vel) return fail("velocity output pointer is NULL but compute_vel is set");
    if (out_dtype != NBE_F32 && out_dtype != NBE_F16) return fail("out_dtype must be NBE_F32 or NBE_F16");
    if (brick_resume(c, 4)) return 1;
    BrickOff off{c};
    if (c->sst.Dz != Dz || c->sst.vel_fac != vel_fac || c->sst.act_scale != c->act_scale)
        return fail("nbe_brick_finish: Dz, vel_fac and the range shift must be those of the nbe_brick_encode call it completes");
    c->bio.recv_lo = recv_lo; c->bio.recv_hi = recv_hi; c->bio.skip_recv_lo = skip_recv_lo; c->bio.skip_recv_hi = skip_recv_hi;
    c->bio.skip_ready = (hipEvent_t)skip_ready_event;
    const int b0 = c->sst.D - 96, S1 = c->sst.H - 2, S2 = c->sst.W - 2;
    const HeadOut ho{disp, vel, out_dtype, b0, S1, S2, 0, 0, 0, Dz, vel_fac};
    if (network_stream(c, c->sst.tin, ho, c->sst.S)) return 1;
    HIPCHK(hipGetLastError());
    return 0;
}

int nbe_check_finite(nbe_ctx* c) {
    if (!c) return fail("null context");
    HIPCHK(hipSetDevice(c->device));
    return check_range(c);
}

int nbe_set_input_range(nbe_ctx* c, float absmax) {
    if (!c) return fail("null context");
    c->preset_absmax = (absmax >= 0.f || std::isnan(absmax)) ? absmax : -1.f;
    return 0;
}

int nbe_query(nbe_ctx* c, int what, double* out) {
    if (!c || !out) return fail("null argument");
    switch (what) {
    case NBE_Q_GAUGE_ACTIVE: *out = c->gauge_active ? 1 : 0; break;
    case NBE_Q_SLAB: *out = c->slab; break;
    case NBE_Q_PERIODIC_YX: *out = c->pyx ? 1 : 0; break;
    case NBE_Q_PERIODIC_Z: *out = c->pz ? 1 : 0; break;
    case NBE_Q_RANGE_SHIFT: *out = std::log2((double)c->act_scale); break;
    case NBE_Q_WORKSPACE_BYTES: *out = (double)c->ws_bytes; break;
    case NBE_Q_HOST_PIPE: *out = c->last_piped ? 1 : 0; break;
    case NBE_Q_GRAPH_REPLAYS: *out = (double)c->graph_replays; break;
    case NBE_Q_PLAN_TILES: *out = (double)c->plan_tiles; break;
    case NBE_Q_PLAN_SHORT_GB: *out = c->plan_short_gb; break;
    default: return fail("nbe_query: unknown item %d", what);
    }
    return 0;
}

// ---- branch probe (test instrumentation) ------------------------------------------------------------------------------
static void probe_free(nbe_ctx* c) {
    (void)hipFree(c->probe.bits); (void)hipFree(c->probe.count);
    c->probe = nbe_ctx::Probe();
}

int nbe_probe_begin(nbe_ctx* c, const int64_t origin[3], int nout) {
    if (!c || !origin) return fail("null argument");
    if (nout < 8 || nout % 8 != 0 || nout > 128) return fail("branch probe: the block edge must be a multiple of 8 in 8..128");
    for (int d = 0; d < 3; ++d) if (origin[d] < 0 || origin[d] % 8 != 0) return fail("branch probe: the block origin must be a non-negative multiple of 8");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    probe_free(c);
    auto& P = c->probe;
    P.nout = nout;
    for (int d = 0; d < 3; ++d) P.p[d] = (int)origin[d];
    // the activation tensors of the cone of an (nout + 96)^3 input, in execution order (core :105-195)
    const int m = c->mid, n0 = nout + 96;
    const int m1 = (n0 - 8) / 2, m2 = (m1 - 4) / 2, m3 = (m2 - 4) / 2, u2 = 2 * (m3 - 4), u1 = 2 * (u2 - 4), u0 = 2 * (u1 - 4);
    struct Row { const char* name; int C, n, level; };
    const Row rows[] = {
        {"conv_l00/conv_0", m, n0 - 2, 0}, {"conv_l00/conv_1", m, n0 - 4, 0}, {"conv_l01/conv_0", m, n0 - 6, 0}, {"conv_l01/conv_1", m, n0 - 8, 0},
        {"down_l0/conv_0", m, m1, 1}, {"conv_l1/conv_0", m, m1 - 2, 1}, {"conv_l1/conv_1", m, m1 - 4, 1},
        {"down_l1/conv_0", m, m2, 2}, {"conv_l2/conv_0", m, m2 - 2, 2}, {"conv_l2/conv_1", m, m2 - 4, 2},
        {"down_l2/conv_0", m, m3, 3}, {"conv_c/conv_0", m, m3 - 2, 3}, {"conv_c/conv_1", m, m3 - 4, 3},
        {"up_r2/conv_0", m, u2, 2}, {"conv_r2/conv_0", 2 * m, u2 - 2, 2}, {"conv_r2/conv_1", m, u2 - 4, 2},
        {"up_r1/conv_0", m, u1, 1}, {"conv_r1/conv_0", 2 * m, u1 - 2, 1}, {"conv_r1/conv_1", m, u1 - 4, 1},
        {"up_r0/conv_0", m, u0, 0}, {"conv_r00/conv_0", 2 * m, u0 - 2, 0}, {"conv_r00/conv_1", m, u0 - 4, 0},
        {"conv_r01/conv_0", m, u0 - 6, 0},
    };
    int64_t off = 0;
    for (const Row& r : rows) {
        nbe_ctx::Probe::Slot sl{r.name, r.C, r.n, (r.n + 31) / 32, r.level, off};
        off += (int64_t)sl.C * sl.n * sl.n * sl.nw;
        P.slots.push_back(sl);
    }
    P.words = off;
    HIPCHK(hipMalloc((void**)&P.bits, off * 4));
    HIPCHK(hipMemset(P.bits, 0, off * 4));
    HIPCHK(hipMalloc((void**)&P.count, P.slots.size() * 4));
    HIPCHK(hipMemset(P.count, 0, P.slots.size() * 4));
    P.on = true; P.tile = false;
    return 0;
}

int nbe_probe_slots(nbe_ctx* c) { return c ? (int)c->probe.slots.size() : 0; }

int nbe_probe_layout(nbe_ctx* c, int slot, char* name, int cap, int dims[3], int64_t* word_offset) {
    if (!c || slot < 0 || slot >= (int)c->probe.slots.size()) return fail("branch probe: slot out of range");
    const auto& S = c->probe.slots[slot];
    if (name && cap > 0) { strncpy(name, S.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (dims) { dims[0] = S.C; dims[1] = S.n; dims[2] = S.nw; }
    if (word_offset) *word_offset = S.off;
    return 0;
}

int nbe_probe_read(nbe_ctx* c, void* words, int64_t nwords) {
    if (!c || !words) return fail("null argument");
    auto& P = c->probe;
    if (!P.on) return fail("branch probe: nbe_probe_begin has not been called");
    if (nwords != P.words) return fail("branch probe: the buffer must hold %lld words", (long long)P.words);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<unsigned> cnt(P.slots.size());
    HIPCHK(hipMemcpy(cnt.data(), P.count, cnt.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < P.slots.size(); ++i) {
        const auto& S = P.slots[i];
        const int64_t want = (int64_t)S.C * S.n * S.n * S.nw;
        if ((int64_t)cnt[i] != want)
            return fail("branch probe: %s was recorded %u times over %lld words (the block must lie inside one tile of the plan; "
                        "brick mode is not probed)", S.name.c_str(), cnt[i], (long long)want);
    }
    HIPCHK(hipMemcpy(words, P.bits, P.words * 4, hipMemcpyDeviceToHost));
    return 0;
}

int nbe_probe_end(nbe_ctx* c) {
    if (!c) return fail("null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    probe_free(c);
    return 0;
}

// ---- test hooks -----------------------------------------------------------------------------------

int nbe_test_modulate(nbe_ctx* c, const float* weight, const float* sw, const float* sb, int cout, int cin, int k,
                      float s0, float s1, float eps, int first_layer, float* w_n, float* dw_tot) {
    if (!c) return fail("null context");
    HIPCHK(hipSetDevice(c->device));
    const size_t nw = (size_t)cout * cin * k * k * k;
    float *dw_ = nullptr, *dsw = nullptr, *dsb = nullptr, *dwn = nullptr, *ddw = nullptr;
    HIPCHK(hipMalloc((void**)&dw_, nw * 4)); HIPCHK(hipMalloc((void**)&dsw, cin * 8)); HIPCHK(hipMalloc((void**)&dsb, cin * 4));
    HIPCHK(hipMalloc((void**)&dwn, nw * 4)); HIPCHK(hipMalloc((void**)&ddw, nw * 4));
    HIPCHK(hipMemcpy(dw_, weight, nw * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsw, sw, cin * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsb, sb, cin * 4, hipMemcpyHostToDevice));
    launch_modulate(dw_, dsw, dsb, cout, cin, k * k * k, s0, s1, eps, first_layer, dwn, dw_tot ? ddw : nullptr, c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(w_n, dwn, nw * 4, hipMemcpyDeviceToHost));
    if (dw_tot) HIPCHK(hipMemcpy(dw_tot, ddw, nw * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dw_); (void)hipFree(dsw); (void)hipFree(dsb); (void)hipFree(dwn); (void)hipFree(ddw);
    return 0;
}

}  // extern "C"

// beta != NULL: the gauged form of a 3x3x3 layer (conv_h3g_kernel / conv_h3w_kernel / their float32 and float16 siblings):
// dx is the tangent in this layer's gauge, dy = W.dx + beta[o] * (W.x); dw is not used
static int test_layer(nbe_ctx* c, int kind, int crop, int flags, const float* x, const float* dx, int cin, int D, int H, int W,
                      const float* w, const float* dw, const float* bias, int cout, const float* res, const float* dres,
                      float* y, float* dy, const float* beta) {
    if (!c || !x || !w || !bias || !y) return fail("null argument");
    if (kind < 0 || kind > 3) return fail("kind must be 0..3");
    if (beta && !(kind == 0 && dx && dy)) return fail("the gauged form belongs to 3x3x3 layers with an input tangent");
    HIPCHK(hipSetDevice(c->device));
    const bool vel = (dw != nullptr || beta != nullptr) && dy != nullptr, has_dx = vel && dx != nullptr;
    const bool saved_vel = c->vel, saved_ga = c->gauge_active, saved_wino = c->wino_ok;
    c->vel = vel;
    c->sst.valid = false;
    const int k = kind == 0 ? 3 : kind == 1 ? 1 : 2;
    int OD, OH, OW;
    if (kind == 0) { OD = D - 2; OH = H - 2; OW = W - 2; }
    else if (kind == 1) { OD = D - 2 * crop; OH = H - 2 * crop; OW = W - 2 * crop; }
    else if (kind == 2) { OD = D / 2; OH = H / 2; OW = W / 2; }
    else { OD = 2 * D; OH = 2 * H; OW = 2 * W; }
    const size_t nin = (size_t)cin * D * H * W, nout = (size_t)cout * OD * OH * OW, nw = (size_t)cout * cin * k * k * k;
    int rc = 0;
    float *dxin = nullptr, *ddx = nullptr, *dwt = nullptr, *ddw = nullptr, *dout = nullptr;
    Layer L;
    char* ws = nullptr;
    do {
        L.cout = cout; L.cin = cin; L.k = k; L.kind = kind;
        PackedW& pw = L.pw;
        pw.mode = kind == 0 ? MODE_FLAT3 : (kind == 2 ? MODE_DOWN : MODE_FLAT1);
        pw.prec = c->prec;
        pw.ni = (prec_is_half(c->prec) || cout > 32) ? 2 : 1; pw.cin = cin; pw.cout = cout;
        pw.cin_pad = roundup(cin, prec_ck(c->prec, pw.mode));
        pw.ctiles = (cout + 32 * pw.ni - 1) / (32 * pw.ni);
        pw.nsets = kind == 3 ? 8 : 1;
        pw.floats = (int64_t)pw.ctiles * 32 * pw.ni * mode_nseg(pw.mode) * mode_taps(pw.mode) * pw.cin_pad / (c->prec == PREC_F16 ? 2 : 1);
#define TCHK(e) if ((e) != hipSuccess) { rc = fail("hip error in nbe_test_layer: %s", hipGetErrorString(hipGetLastError())); break; }
        TCHK(hipMalloc((void**)&pw.w, pw.floats * pw.nsets * 4));
        if (vel) TCHK(hipMalloc((void**)&pw.dw, pw.floats * pw.nsets * 4));
        if (prec_is_half(c->prec) && kind == 0 && !has_dx && cin <= 3 && cout <= 64)     // as conv_l00/conv_0: stem_h3_kernel
            TCHK(hipMalloc((void**)&pw.stem, 4 * 3 * 4 * 64 * 16));
        const int nb = pw.ctiles * 32 * pw.ni;
        TCHK(hipMalloc((void**)&pw.bias, nb * 4)); TCHK(hipMemset(pw.bias, 0, nb * 4));
        TCHK(hipMemcpy(pw.bias, bias, cout * 4, hipMemcpyHostToDevice));
        TCHK(hipMalloc((void**)&dwt, nw * 4)); TCHK(hipMemcpy(dwt, w, nw * 4, hipMemcpyHostToDevice));
        launch_pack(dwt, cout, cin, kind, pw, pw.w, c->stream);
        if (vel && !beta) { TCHK(hipMalloc((void**)&ddw, nw * 4)); TCHK(hipMemcpy(ddw, dw, nw * 4, hipMemcpyHostToDevice));
                            launch_pack(ddw, cout, cin, kind, pw, pw.dw, c->stream); }
        if (beta) {
            const size_t nbt = (size_t)pw.ctiles * 32 * pw.ni + 64;
            TCHK(hipMalloc((void**)&L.beta, nbt * 4)); TCHK(hipMemset(L.beta, 0, nbt * 4));
            TCHK(hipMemcpy(L.beta, beta, cout * 4, hipMemcpyHostToDevice));
            L.g6 = true; c->gauge_active = true; c->wino_ok = false;
            if ((c->prec == PREC_F16X3 && pw.cin_pad / 16 <= 8) || wino_f16_layer(c->prec, true, pw.cin_pad)) {
                TCHK(hipMalloc((void**)&pw.ww, pw.floats * 4 / 3 * 4));
                if (!c->wino_flag) TCHK(hipMalloc((void**)&c->wino_flag, 4));
                TCHK(hipMemsetAsync(c->wino_flag, 0, 4, c->stream));
                launch_pack_h3w(dwt, cout, cin, pw.cin_pad, pw.ctiles, pw.ww, c->wino_flag, c->stream, c->prec);
                int bad = 0;
                TCHK(hipMemcpyAsync(&bad, c->wino_flag, 4, hipMemcpyDeviceToHost, c->stream));
                TCHK(hipStreamSynchronize(c->stream));
                c->wino_ok = bad == 0;
            }
        }
        if (!vel && kind == 0 && c->prec == PREC_F16X3 && cin > 3 && pw.cin_pad / 16 <= 8) {
            // displacement only: conv_h3w_kernel<false, NOVEL> when the output has an even number of planes (as run_conv decides)
            TCHK(hipMalloc((void**)&pw.ww, pw.floats * 4 / 3 * 4));
            if (!c->wino_flag) TCHK(hipMalloc((void**)&c->wino_flag, 4));
            TCHK(hipMemsetAsync(c->wino_flag, 0, 4, c->stream));
            launch_pack_h3w(dwt, cout, cin, pw.cin_pad, pw.ctiles, pw.ww, c->wino_flag, c->stream);
            int bad = 0;
            TCHK(hipMemcpyAsync(&bad, c->wino_flag, 4, hipMemcpyDeviceToHost, c->stream));
            TCHK(hipStreamSynchronize(c->stream));
            c->wino_ok = bad == 0;
        }
        TCHK(hipMalloc((void**)&dxin, nin * 4)); TCHK(hipMemcpy(dxin, x, nin * 4, hipMemcpyHostToDevice));
        if (has_dx) { TCHK(hipMalloc((void**)&ddx, nin * 4)); TCHK(hipMemcpy(ddx, dx, nin * 4, hipMemcpyHostToDevice)); }
        TCHK(hipMalloc((void**)&dout, nout * 4));
        // private workspace: input, output, residual planes
        auto mk = [&](int C, int d, int h, int wd, int64_t* bytes) {
            Planes p; p.G = planes_for(C, c->prec); p.D = d; p.H = h; p.W = wd; p.pstride = (p.vox() + 63) & ~int64_t(63);
            *bytes = (int64_t)p.G * p.pstride * 16; return p; };
        int64_t bi, bo;
        Planes pin = mk(cin, D, H, W, &bi), pout = mk(cout, OD, OH, OW, &bo), pres = pout;
        const int64_t tot = 2 * bi + 4 * bo + ((int64_t)2 * H * W + 2 * W + 1024) * 16;
        // NBE_TEST_ADDR_BIT31 = 0 / 1 places the tensors where bit 31 of their addresses is clear / set.  The global -> LDS
        // DMA of the 16x16x32 kernels splits its wave-uniform base into two 32-bit halves (readfirstlane) and joins them
        // again (dma16s); a join that sign-extends the low half is wrong exactly when that bit is set -- the memory access
        // fault at 0xffffbf6e4000 of round 1 -- and right for every other address.
        char* wb = nullptr;
        if (const char* e = getenv("NBE_TEST_ADDR_BIT31")) {
            const uint64_t two = 1ull << 31, want = atoi(e) ? 1 : 0;
            if ((uint64_t)tot >= two) { rc = fail("NBE_TEST_ADDR_BIT31 needs a test tensor below 2 GiB"); break; }
            TCHK(hipMalloc((void**)&ws, tot + 2 * two));
            uint64_t b = ((uint64_t)ws + two - 1) & ~(two - 1);
            if (((b >> 31) & 1) != want) b += two;
            wb = (char*)b;
        } else {
            TCHK(hipMalloc((void**)&ws, tot));
            wb = ws;
        }
        TCHK(hipMemsetAsync(wb, 0, tot, c->stream));
        pin.x = (float*)wb; pin.dx = (float*)(wb + bi);
        pout.x = (float*)(wb + 2 * bi); pout.dx = (float*)(wb + 2 * bi + bo);
        pres.x = (float*)(wb + 2 * bi + 2 * bo); pres.dx = (float*)(wb + 2 * bi + 3 * bo);
        launch_to_planes(dxin, cin, pin, false, 1.0f, c->prec, c->stream);
        if (has_dx) launch_to_planes(ddx, cin, pin, true, 1.0f, c->prec, c->stream);
        if (flags & F_RES) {
            if (!res) { rc = fail("residual flag set but res is NULL"); break; }
            TCHK(hipMemcpyAsync(dout, res, nout * 4, hipMemcpyHostToDevice, c->stream));
            launch_to_planes(dout, cout, pres, false, 1.0f, c->prec, c->stream);
            if (vel && dres) { TCHK(hipStreamSynchronize(c->stream)); TCHK(hipMemcpyAsync(dout, dres, nout * 4, hipMemcpyHostToDevice, c->stream));
                               launch_to_planes(dout, cout, pres, true, 1.0f, c->prec, c->stream); }
            TCHK(hipStreamSynchronize(c->stream));
        }
        ConvLaunch cl; cl.in = pin; cl.out = pout; cl.res = pres; cl.flags = flags;
        if (kind == 0) { cl.Dv = OD; cl.Hv = OH; cl.Wv = OW; rc = run_conv(c, L, cl, has_dx); }
        else if (kind == 1) { cl.in_off = ((int64_t)crop * H + crop) * W + crop; cl.Dv = OD; cl.Hv = OH; cl.Wv = OW; rc = run_conv(c, L, cl, has_dx); }
        else if (kind == 2) { cl.Dv = OD; cl.Hv = OH; cl.Wv = OW; rc = run_conv(c, L, cl, has_dx); }
        else {
            // as upblock(): one launch for all eight parities where up_h3_kernel applies
            const bool up8 = prec_is_half(c->prec) && (!vel || has_dx) && pw.cin_pad <= 64 && !(getenv("NBE_UP8") && atoi(getenv("NBE_UP8")) == 0);
            for (int p = 0; p < (up8 ? 1 : 8) && !rc; ++p) {
                ConvLaunch u = cl; u.Dv = D; u.Hv = H; u.Wv = W; u.osz = 2; u.oz = (p >> 2) & 1; u.oy = (p >> 1) & 1; u.ox = p & 1;
                u.set = up8 ? -1 : p;
                rc = run_conv(c, L, u, has_dx);
            }
        }
        if (rc) break;
        launch_from_planes(pout, false, cout, dout, c->prec, c->stream);
        TCHK(hipStreamSynchronize(c->stream));
        TCHK(hipMemcpy(y, dout, nout * 4, hipMemcpyDeviceToHost));
        if (vel) {
            launch_from_planes(pout, true, cout, dout, c->prec, c->stream);
            TCHK(hipStreamSynchronize(c->stream));
            TCHK(hipMemcpy(dy, dout, nout * 4, hipMemcpyDeviceToHost));
        }
        TCHK(hipGetLastError());
#undef TCHK
    } while (0);
    c->vel = saved_vel; c->gauge_active = saved_ga; c->wino_ok = saved_wino;
    (void)hipFree(dxin); (void)hipFree(ddx); (void)hipFree(dwt); (void)hipFree(ddw); (void)hipFree(dout); (void)hipFree(ws);
    (void)hipFree(L.pw.w); (void)hipFree(L.pw.dw); (void)hipFree(L.pw.bias); (void)hipFree(L.pw.stem); (void)hipFree(L.pw.ww); (void)hipFree(L.beta);
    return rc;
}

extern "C" {

int nbe_test_layer(nbe_ctx* c, int kind, int crop, int flags, const float* x, const float* dx, int cin, int D, int H, int W,
                   const float* w, const float* dw, const float* bias, int cout, const float* res, const float* dres,
                   float* y, float* dy) {
    return test_layer(c, kind, crop, flags, x, dx, cin, D, H, W, w, dw, bias, cout, res, dres, y, dy, nullptr);
}

int nbe_test_layer_gauged(nbe_ctx* c, int flags, const float* x, const float* dx, int cin, int D, int H, int W,
                          const float* w, const float* beta, const float* bias, int cout, float* y, float* dy) {
    if (!beta) return fail("null argument");
    return test_layer(c, 0, 0, flags, x, dx, cin, D, H, W, w, nullptr, bias, cout, nullptr, nullptr, y, dy, beta);
}

int nbe_test_layer_gauged_res(nbe_ctx* c, int flags, const float* x, const float* dx, int cin, int D, int H, int W,
                              const float* w, const float* beta, const float* bias, int cout, const float* res, const float* dres,
                              float* y, float* dy) {
    if (!beta) return fail("null argument");
    return test_layer(c, 0, 0, flags, x, dx, cin, D, H, W, w, nullptr, bias, cout, res, dres, y, dy, beta);
}

int nbe_profile_enable(nbe_ctx* c, int on) { if (!c) return fail("null context"); prof_collect(c); c->prof = on != 0; return 0; }
int nbe_profile_reset(nbe_ctx* c) { if (!c) return fail("null context"); prof_collect(c); c->prof_entries.clear(); return 0; }
int nbe_profile_count(nbe_ctx* c) { if (!c) return 0; prof_collect(c); return (int)c->prof_entries.size(); }
int nbe_profile_entry(nbe_ctx* c, int i, char* name, int cap, double* ms, int64_t* launches, double* flops) {
    if (!c || i < 0 || i >= (int)c->prof_entries.size()) return fail("profile entry out of range");
    const ProfEntry& e = c->prof_entries[i];
    if (name && cap > 0) { strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (ms) *ms = e.ms; if (launches) *launches = e.launches; if (flops) *flops = e.flops;
    return 0;
}
int64_t nbe_workspace_bytes(nbe_ctx* c) { return c ? c->ws_bytes : 0; }

int nbe_debug_phase_cycles(nbe_ctx* c, double* out16) {
    if (!c || !out16) return fail("null argument");
    h3q_read_stamps(out16, c->stream);
    return 0;
}

}  // extern "C"
