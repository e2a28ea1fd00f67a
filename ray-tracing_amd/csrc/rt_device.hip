// rt_device.hip -- the gfx950 render kernel.
//
// Compiled twice into one library:
//   strict flavour (default):  -ffp-contract=off, correctly rounded / and sqrt, the
//       reference shader's transcendental kernels restated with explicit fma.  Every FP32
//       operation is the one llvmpipe executes for fragment_shader.glsl, in its order, so the
//       frame is bit-identical to the reference's.
//   fast flavour (-DRT_FAST):  -ffp-contract=fast plus v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_cos.
// Both are the same source; the flavour only changes the primitives in the "numerics" block.
//
// Design (wave64, VALU-bound, no MFMA: there is no dense contraction in this workload):
//   * one lane = one pixel at a time; a wave owns an 8x8 pixel tile.  Small scenes: one tile per
//     wave and launch, one wave per workgroup.  Large scenes and deep stacks: persistent waves that
//     take tiles from a work queue (REFILL below).
//   * the scene is staged once per workgroup in LDS, split so that the per-object hot loop
//     reads one 16-byte record (centre + bounding radius^2) as an LDS broadcast; torus
//     frames, kinds and materials sit behind the bounding test / the closest-hit decision.
//     Scenes of 40+ objects are traced through a sphere hierarchy that returns exactly the
//     per-object loop's hits, with the member tests and exact intersections of a wave's 64 rays
//     pooled in LDS and dealt out evenly over the lanes (rt_intersect.h).
//   * trace() keeps only (t, id) of the closest hit; position, normal and material are
//     derived once for the winner (the reference builds a full HitInfo per candidate,
//     fragment_shader.glsl:416-425; only the winner's is ever read).
//   * the reference's explicit ray stack (fragment_shader.glsl:547-564) is "current ray +
//     pending refraction rays": the reflection child is always popped right after it is
//     pushed, so it never needs a slot.  The pending slots live in LDS, in a global workspace
//     or in registers (PEND below).  Visit order and the silent drop rule (stack_ptr <
//     STACK_SIZE, :615/:632) are preserved exactly.
//   * path rays and shadow rays go through ONE trace site: each lane is a small state
//     machine (PATH / SHADOW), so lanes shading and lanes casting shadow rays share the
//     wave's trace iterations instead of serialising two inlined copies of trace().
//   * lanes of a wave are kept in lock-step wherever path lengths allow it (whole tiles, samples
//     started together): when lanes drift apart every code block of the loop runs on every
//     iteration, which costs more than the idle lanes it saves -- except at depth 16 on large
//     scenes, where per-lane pixel refill wins.
//   * tiles are dispatched down the columns of the image, the columns from the centre outwards
//     (grid of the one-pixel-per-lane kernels, tile queue of the persistent ones): a frame cannot
//     finish before its deepest tile, rows of an image are alike and columns cross everything, so
//     this order keeps cheap and deep tiles mixed on every SIMD and starts the deep ones early.
// The arithmetic primitives are in rt_numerics.h, intersection and trace in rt_intersect.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#include "rt_types.h"

#ifdef RT_FAST
#define RT_NS rt_fast
#else
#define RT_NS rt_strict
#endif

#define DEV __device__ __forceinline__

#include "rt_numerics.h"
#include "rt_intersect.h"

namespace RT_NS {

// ------------------------------------------------------------------ pending-ray slots
template <int CAP> struct Pending {
    float ox[CAP], oy[CAP], oz[CAP], dx[CAP], dy[CAP], dz[CAP], tx[CAP], ty[CAP], tz[CAP], ior[CAP];
    int depth[CAP];
};

#define SEL_STORE(arr, val) arr[k] = (k == slot) ? (val) : arr[k]
#define SEL_LOAD(dst, arr) dst = (k == slot) ? arr[k] : dst

// ------------------------------------------------------------------ the kernel
enum { ST_PATH = 0, ST_SHADOW = 1 };

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 2
#endif
#ifndef RT_WAVES_PER_EU_HIER
#define RT_WAVES_PER_EU_HIER 3
#endif
// single-sample hierarchy kernels: 128 VGPRs (two dwords spilled) for a fourth wave per SIMD, measured
// 64 tori 4K 1.76 -> 1.58 ms, 256 spheres 4K depth 16 351 -> 332 ms; the multi-sample ones would spill
// 23 dwords and lose 4 %
#ifndef RT_WAVES_PER_EU_HIER1
#define RT_WAVES_PER_EU_HIER1 4
#endif
#ifndef RT_WAVES_PER_EU_LDSP
#define RT_WAVES_PER_EU_LDSP 4
#endif
// Template flags of the kernel:
// MULTI: more than one sample per pixel (sample loop + accumulators); the single-sample
// kernel carries neither.
// HIER: trace through the sphere hierarchy (large scenes); only the hierarchy sections and the
// waves' work lists (RT_FLAT; per-lane candidate queues without it) are in LDS, the rest of the
// scene is read from global memory.
// REFILL: persistent waves behind a work queue.  The frame is a queue of 8x8 tiles in
// RT_SUBQUEUES interleaved sub-queues; a wave takes a tile, and whenever lanes finish their
// pixel they are handed the next unassigned pixels of the wave's current tile (ballot + prefix
// count), a new tile being taken when it runs out.  Path-tree sizes differ by orders of
// magnitude between neighbouring pixels at depth 16, so without this most lanes of a wave
// wait for its longest pixel.  With kp.whole_tiles the pixels are handed out only when the
// whole wave is idle: tile after tile with the lanes in lock-step (small scenes).
// TOPPOOL (with POOL): the loop over the kept top nodes is pooled as well (large hierarchies).
// POOL (hierarchy kernels): the member tests and exact intersections of a wave's rays are pooled
// (trace_hierarchy_flat); without it every lane walks its own lists (trace_hierarchy), which has less
// fixed cost per trace and is what small scenes of cheap objects run.
// DEEP (persistent hierarchy kernels): the code for frames of deep path trees -- a wave's pixels
// come from 64 different tiles (kp.spread_shift / kp.spread_stride), and a wave with at most
// kp.coop_rays rays left traces them one at a time with all lanes (trace_cooperative).
// PILOT (persistent hierarchy kernels): the launch that ranks the pixels of a deep frame by cost.  A lane's
// "pixel" is the top-left pixel of a 2x2 pixel cell; instead of a colour the lane stores the number of rays
// the pixel's tree took (to the small depth the host sets), saturated to a byte (kp.pilot_cost).
// PEND: where the pending-ray slots live: PEND_REG registers (CAP of them, compare/select
// addressing), PEND_LDS ([slot][field][thread] after the scene), PEND_GLOBAL (same layout in a
// per-workgroup slice of a global workspace; persistent-wave kernels only, whose grid is
// bounded by the number of resident workgroups), PEND_HYBRID (the first three slots in LDS,
// deeper ones in the workspace: most paths never go deeper, and the LDS footprint stays that
// of the three-slot kernel).
enum { PEND_REG = 0, PEND_LDS = 1, PEND_GLOBAL = 2, PEND_HYBRID = 3 };
#define RT_HYBRID_LDS_SLOTS 3      // PEND_HYBRID: slots 0..2 in LDS, the rest in the global workspace
// Threads per workgroup.  The one-pixel-per-lane kernels with LDS slots run one wave (one 8x8 tile)
// per workgroup: a workgroup keeps its LDS and its wave slots until its last wave ends, and ray
// trees differ a lot between neighbouring tiles, so the three waves that finish first would leave
// their SIMD slots empty until the fourth does.
// 1: the hierarchy kernels pool the member tests and exact intersections of a wave's 64 rays
// (trace_hierarchy_flat); 0: every lane walks its own lists (trace_hierarchy)
#ifndef RT_FLAT
#define RT_FLAT 1
#endif
#ifndef RT_WG_ONE
#define RT_WG_ONE 64
#endif
template <int PEND, bool DEBUG, bool REFILL> constexpr int rt_wg_size() { return (PEND == PEND_LDS && !DEBUG && !REFILL) ? RT_WG_ONE : 256; }
template <int CAP, bool DEBUG, int PEND, bool MULTI, bool HIER, bool REFILL, bool DEEP = false, bool POOL = true, bool TOPPOOL = false, bool PILOT = false, bool WHOLE = false>
__global__ __launch_bounds__((rt_wg_size<PEND, DEBUG, REFILL>()), (PEND == PEND_LDS || PEND == PEND_HYBRID) ? RT_WAVES_PER_EU_LDSP : (HIER && (CAP <= 7 || PEND == PEND_GLOBAL)) ? (MULTI ? RT_WAVES_PER_EU_HIER : RT_WAVES_PER_EU_HIER1) : RT_WAVES_PER_EU)
void rt_kernel(KParams kp, const float *__restrict__ scene, void *__restrict__ out,
               unsigned long long *__restrict__ ev_out, float *__restrict__ cap_out)
{
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    constexpr int WG = rt_wg_size<PEND, DEBUG, REFILL>();
    const SceneLayout L = kp.scene;
    // HIER keeps only the hierarchy sections in LDS (they are what every ray reads many
    // times); bound records, materials and torus frames are read a few times per ray and
    // come from global memory through the caches, which leaves LDS room for several
    // workgroups per CU however large the scene is.  WHOLE (the host's choice for scenes whose
    // whole blob fits next to the work lists in a quarter of the CU's LDS): everything in LDS --
    // the loads of the exact intersections and of shading then have LDS latency.
    constexpr bool SPLIT = HIER && !WHOLE;      // hierarchy in LDS, the rest in global memory
    const int lds_first = SPLIT ? L.off_top : 0;
    const int lds_dwords = L.total_dwords - lds_first;
    {
        const float4 *src = reinterpret_cast<const float4 *>(scene + lds_first);
        const int nq = lds_dwords >> 2;
        for (int i = threadIdx.x; i < nq; i += WG) lds4[i] = src[i];
    }
    const float *sc = SPLIT ? scene : lds;      // bound records, materials, torus frames, kinds
    const float *hier = lds + ((HIER && WHOLE) ? L.off_top : 0);        // hierarchy sections (HIER)
    (void)hier;
    Counters<DEBUG> cnt;
    cnt.lds_ev = reinterpret_cast<unsigned int *>(lds + lds_dwords);
    if constexpr (DEBUG) cnt.wave_mode = kp.wave_profile;
    if (DEBUG) { if (threadIdx.x < RT_COUNTER_SLOTS) cnt.lds_ev[threadIdx.x] = 0u; }
    __syncthreads();

    // ---- which pixel
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int px, packed_row, gy;     // pixel column, row inside this part's buffer, GL row (0 = bottom)
    bool active;
    if (!REFILL) {
        // Dispatch order of the 16x16 tiles (the hardware walks blockIdx.x fastest): down the
        // columns of the image, the columns taken from the centre outwards.  Rows of an image are
        // alike (all sky, all ground) and a column crosses everything, so in raster order the
        // deep tiles of the object rows run together and the last ones finish alone; down the
        // columns every stretch of the dispatch sequence holds the frame's mix of cheap and deep
        // tiles, and the deep ones (a centred subject) are under way early.  The image does not
        // depend on the order.
        const int tiles_x = (int)gridDim.y;
        const int k = (int)blockIdx.y, mid = tiles_x >> 1;
        const int bx = (k & 1) ? (mid - 1 - (k >> 1)) : (mid + (k >> 1));
        int by = (int)blockIdx.x, in_band;
        if (WG == 64) {
            // one 8x8 tile per workgroup: two tile rows per 16-row band
            px = bx * 8 + (lane & 7);
            in_band = (by & 1) * 8 + (lane >> 3);
            by >>= 1;
        } else {
            px = bx * 16 + (wave & 1) * 8 + (lane & 7);
            in_band = (wave >> 1) * 8 + (lane >> 3);
        }
        const int band = kp.part + by * kp.nparts;
        const int row = band * 16 + in_band;             // output row
        packed_row = by * 16 + in_band;
        active = px < kp.w && row < kp.h;
        gy = kp.bottom_up ? row : (kp.h - 1 - row);
    } else {
        px = 0; packed_row = 0; gy = 0; active = false;
    }

    // SYNC: with several samples per pixel and one pixel per lane, a lane that has finished a
    // sample waits for the rest of its wave before it starts the next one: lanes that run ahead
    // leave lock-step and then every code block of the loop executes on every iteration
    constexpr bool SYNC = MULTI && !REFILL && !DEBUG;
    bool parked = false;
    const int N = MULTI ? kp.spp : 1;
    const int nsamples = N * N;
    const float inv_spp = 1.0f / (float)nsamples;

    // ---- per-lane state
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 1, dz = 0;  // ray being traced
    float thx = 1, thy = 1, thz = 1, cur_ior = 1;          // path segment: throughput, medium
    int depth = 0;
    float fr = 0, fg = 0, fb = 0;                          // colour of the current sample
    float accr = 0, accg = 0, accb = 0;                    // sum over samples
    int sample = 0;
    int np = 0;                                            // pending refraction rays
    constexpr bool LDSP = PEND == PEND_LDS || PEND == PEND_HYBRID;
    constexpr bool MEMP = PEND != PEND_REG;             // slots in memory, not registers
    Pending<MEMP ? 1 : CAP> pend;
#pragma unroll
    for (int k = 0; k < (MEMP ? 1 : CAP); ++k) {
        pend.ox[k] = pend.oy[k] = pend.oz[k] = 0; pend.dx[k] = pend.dy[k] = pend.dz[k] = 0;
        pend.tx[k] = pend.ty[k] = pend.tz[k] = 0; pend.ior[k] = 1; pend.depth[k] = 0;
    }
    // LDS pending slots start after the scene (and the debug counters), 16-byte aligned
    float *pend_lds = lds + ((lds_dwords + (DEBUG ? RT_COUNTER_SLOTS : 0) + 3) & ~3) + threadIdx.x;
    float *pend_glb = kp.pend_ws + (size_t)blockIdx.x * ((size_t)kp.pend_ws_slots * 11 * 256) + threadIdx.x;
    float *pend_mem = PEND == PEND_GLOBAL ? pend_glb : pend_lds;
    (void)LDSP;
    // candidate queue of the hierarchy walk: [entry][thread], 16 bits each (never together with LDSP)
    unsigned short *queue = reinterpret_cast<unsigned short *>(lds + ((lds_dwords + (DEBUG ? RT_COUNTER_SLOTS : 0) + 3) & ~3)) + threadIdx.x;
    // pooled hierarchy trace: RT_FLAT_DWORDS of LDS per wave, in the same region
    // (the wave number through readfirstlane: the compiler then keeps the wave's LDS base in a scalar register)
    float *flat_lds = lds + ((lds_dwords + (DEBUG ? RT_COUNTER_SLOTS : 0) + 3) & ~3) +
                      __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * RT_FLAT_DWORDS(TOPPOOL);
    (void)flat_lds;
    int state = ST_PATH;
    // shading context carried across the shadow hops of one hit
    float hx = 0, hy = 0, hz = 0;            // hit position
    float nx = 0, ny = 1, nz = 0;            // geometric normal
    float pdx = 0, pdy = 0, pdz = 0;         // direction of the path ray that hit
    float mr = 0, mg = 0, mb = 0, m_refl = 0, m_transp = 0;
    float osign = 1, dni = 0, eta = 1, n2 = 1, F = 0, local = 0;
    float light_dist = 0, travelled = 0, filr = 1, filg = 1, filb = 1;
    float stx = 0, sty = 0, stz = 0;         // shadow step = L * 0.001
    int hop = 0;
    // debug capture
    int seg = 0, cap_nseg = 0; float cap_id = -2.0f, cap_t = 0.0f, cap_depth = 0.0f;

    // stratified sample positions: offset of sample column/row i from the pixel centre,
    // (i + 0.5)/N - 0.5 rounded once to binary32 (host-computed table, kp.sample_off)
    int si = 0, sj = 0;
    auto start_sample = [&]() {
        // [census: start_sample @ START]
        // main() (:650-668): pixel centre + stratified offset
        float fx = (float)px + 0.5f, fy = (float)gy + 0.5f;
        if (MULTI) { fx = fx + kp.sample_off[si]; fy = fy + kp.sample_off[sj]; }
        const float nx = fx * 2.0f - kp.res_w, ny = fy * 2.0f - kp.res_h;
        float uvx, uvy;
        if (!MULTI && kp.uv_by_reciprocal) {
            // pixel centres only: the host has run this sequence on every numerator of the frame size
            uvx = div_by_reciprocal(nx, kp.res_h, kp.inv_res_h);
            uvy = div_by_reciprocal(ny, kp.res_h, kp.inv_res_h);
        } else {
            uvx = rt_div(nx, kp.res_h);
            uvy = rt_div(ny, kp.res_h);
        }
        float rx = (uvx * kp.right[0] + uvy * kp.up[0]) + kp.focal * kp.fwd[0];
        float ry = (uvx * kp.right[1] + uvy * kp.up[1]) + kp.focal * kp.fwd[1];
        float rz = (uvx * kp.right[2] + uvy * kp.up[2]) + kp.focal * kp.fwd[2];
        float rs = rt_rsq(dot3(rx, ry, rz, rx, ry, rz));
        dx = rx * rs; dy = ry * rs; dz = rz * rs;
        ox = kp.eye[0]; oy = kp.eye[1]; oz = kp.eye[2];
        thx = thy = thz = 1.0f; cur_ior = 1.0f; depth = 0;
        fr = fg = fb = 0.0f; np = 0; state = ST_PATH; seg = 0;
        CNT(EV_PIXEL_PATH);
        WCNT(WP_START);
    };
    // Area light (MULTI kernels, kp.light_grid = G > 1; not a reference feature): the point light
    // becomes a horizontal square sampled on a G x G grid.  Per shaded hit the samples are visited
    // in order (column index fastest), each with its own shadow hops; the light terms of the
    // samples that are not fully blocked are summed from 0 and their mean joins the ambient term.
    int li = 0, lj = 0;                      // light sample column / row
    float lsr = 0, lsg = 0, lsb = 0;         // sum of the light terms so far
    const bool area_light = MULTI && kp.light_grid > 1;
    // calculate_light_attenuation set-up (:443-450) towards the current light sample
    auto setup_shadow = [&]() {
        // [census: setup_shadow @ LOCAL]
        CNT(EV_ATTEN_FIXED);
        float qx = kp.light[0], qz = kp.light[2];
        if (area_light) { qx = qx + kp.light_off[li]; qz = qz + kp.light_off[lj]; }
        float lx = qx - hx, ly = kp.light[1] - hy, lz = qz - hz;
        float d2 = dot3(lx, ly, lz, lx, ly, lz);
        light_dist = rt_sqrt(d2);
        float rs = rt_rsq(d2);
        dx = lx * rs; dy = ly * rs; dz = lz * rs;      // shadow direction L
        stx = dx * 0.001f; sty = dy * 0.001f; stz = dz * 0.001f;
        ox = hx + stx; oy = hy + sty; oz = hz + stz;
        filr = filg = filb = 1.0f; travelled = 0.0f; hop = 0;
    };
    // RGBA8 UNORM conversion of the framebuffer write (:674): clamp, *255, round to even
    auto write_pixel = [&]() {
        // [census: write_pixel @ 1]
        float cr = fr, cg = fg, cb = fb;
        if (MULTI) { cr = accr * inv_spp; cg = accg * inv_spp; cb = accb * inv_spp; }
        size_t o = (size_t)packed_row * kp.w + px;
        if (kp.out_mode == 1) {
            reinterpret_cast<float4 *>(out)[o] = make_float4(cr, cg, cb, 1.0f);
        } else {
            // clamp to [0, 1] with NaN -> 0 (v_med3_f32 returns the minimum of the three when one is
            // NaN), scale, and round to nearest even by adding 2^23: the sum's low mantissa byte is
            // RN(s * 255)
            auto u8 = [](float c) -> uint32_t {
                const float s = __builtin_amdgcn_fmed3f(c, 0.0f, 1.0f);
                return __float_as_uint(s * 255.0f + 8388608.0f) & 0xffu;
            };
            const uint32_t r8 = u8(cr), g8 = u8(cg), b8 = u8(cb);
            if (kp.out_mode == 2) {
                uint8_t *b = reinterpret_cast<uint8_t *>(out) + o * 3;
                b[0] = (uint8_t)r8; b[1] = (uint8_t)g8; b[2] = (uint8_t)b8;
            } else {
                reinterpret_cast<uint32_t *>(out)[o] = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
            }
        }
    };

    // [census: loop_control @ ITER]
#ifdef RT_DEV_SWITCHES
    const unsigned long long dev_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dev_trips = 0, dev_pixels = 0;
#endif
    const bool in_frame = active;       // fixed pixel per lane (without REFILL)
    // wave-uniform queue state (REFILL): the tile being handed out and its next unassigned pixel
    // (the pilot launch walks the grid of 2x2 pixel cells: half as many columns and rows)
    const int tiles_x8 = PILOT ? (kp.cells_x + 7) >> 3 : (kp.w + 7) >> 3;
    const int tile_rows8 = PILOT ? kp.my_bands : kp.my_bands * 2;
    const bool ordered = DEEP && !PILOT && kp.order != nullptr;     // pixels handed out in the pilot's order
    // (whole blocks of kp.order_deal entries: the last block's entries reach past the end of the order and stop there)
    const int ntiles = ordered ? ((kp.order_cells + 16 * kp.order_deal - 1) / (16 * kp.order_deal)) * kp.order_deal : tiles_x8 * tile_rows8;
    int nrays_pixel = 0;                                            // PILOT: rays of the current pixel so far
    (void)nrays_pixel;
    int pool_tile = -1, pool_next = 64;
    int sq = (int)((blockIdx.x * 4u + (unsigned int)wave) % RT_SUBQUEUES), sq_tried = 0;
    bool exhausted = false;
    if (!REFILL && active) start_sample();

    cnt.tick0();
    for (;;) {
        cnt.tick(HS_T_SHADE);
        if (!REFILL) {
            if (SYNC) {
                // lanes start sample k+1 together, once the whole wave has finished sample k
                if (__ballot(active && !parked) == 0ull) {
                    if (__ballot(active) == 0ull) break;
                    if (active) { start_sample(); parked = false; }
                }
            } else if (!active) break;
        } else {
            const unsigned long long need = __ballot(!active);
            // whole_tiles: pixels are handed out only when the whole wave is idle, i.e. a wave
            // renders tile after tile with its lanes in lock-step (small scenes)
            if (need != 0ull && (!kp.whole_tiles || need == ~0ull) && (pool_next < 64 || !exhausted)) {
                const int nneed = __popcll(need);
                const int my = __builtin_amdgcn_mbcnt_hi((unsigned int)(need >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned int)need, 0u));
                const bool cross = pool_next + nneed > 64;
                int new_tile = -1;
                if (cross && !exhausted) {
                    // The queue is RT_SUBQUEUES interleaved sub-queues (sub-queue q holds the tiles
                    // t = q mod RT_SUBQUEUES of the column-major sequence), each behind its own counter: one
                    // counter for the whole GPU sustains only ~1e8 operations per second, which
                    // bounded frames of cheap tiles; interleaving keeps all waves in the same
                    // region of the frame.  A wave drains the sub-queue it starts on, then moves
                    // on to the next one that still has tiles.
                    bool got = false;
                    while (sq_tried < RT_SUBQUEUES) {
                        const int cnt = (ntiles - sq + RT_SUBQUEUES - 1) / RT_SUBQUEUES;   // tiles of sub-queue sq
                        if (cnt > 0) {
                            unsigned int tk = 0;
                            if (lane == 0) tk = atomicAdd(kp.work_counter + sq * RT_SUBQUEUE_STRIDE, 1u);
                            tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
                            if (tk < (unsigned int)cnt) { new_tile = (int)tk * RT_SUBQUEUES + sq; got = true; break; }
                        }
                        sq = sq + 1 == RT_SUBQUEUES ? 0 : sq + 1;
                        sq_tried++;
                    }
                    if (!got) exhausted = true;
                }
                if (!active) {
                    int idx = pool_next + my;
                    int tile = idx < 64 ? pool_tile : new_tile;
                    idx &= 63;
                    if (DEEP && ordered) {
                        // Queue entry `tile` = 16 cells of the cost order, four lanes (2x2 pixels) per cell.  The
                        // order is dealt out like cards: the entries come in blocks of kp.order_deal (one entry
                        // per persistent wave), and inside a block entry e takes the cells e, e + deal,
                        // e + 2 deal, ... of the block's stretch of the order -- every wave starts with the same
                        // mix of the most expensive pixels instead of the first wave with the 64 deepest ones.
                        const int blk = tile / kp.order_deal, eo = tile - blk * kp.order_deal;
                        const int ci = (blk * 16 + (idx >> 2)) * kp.order_deal + eo;
                        if (tile >= 0 && ci < kp.order_cells) {
                            const unsigned int cell = kp.order[ci];
                            px = (int)(cell & 0xffffu) * 2 + (idx & 1);
                            packed_row = (int)(cell >> 16) * 2 + ((idx >> 1) & 1);
                            const int row = (kp.part + (packed_row >> 4) * kp.nparts) * 16 + (packed_row & 15);
                            if (px < kp.w && row < kp.h) {
                                gy = kp.bottom_up ? row : (kp.h - 1 - row);
                                if (MULTI) { accr = accg = accb = 0.0f; sample = 0; si = 0; sj = 0; }
                                start_sample();
                                active = true;
                            }
                        }
                    } else if (tile >= 0) {
                        // Spread: queue entry `tile` stands for pixel groups (1 << spread_shift lanes each) of
                        // 64 >> spread_shift different tiles, kp.spread_stride apart in the tile sequence,
                        // so that a cluster of deep pixels ends up a few lanes each in many waves.  Slot idx
                        // of entry v is slot idx of tile v + (idx >> shift) * stride (mod ntiles): for
                        // every idx a rotation of the tile sequence, hence every pixel exactly once.
                        if (DEEP && kp.spread_shift < 6) {
                            tile += (idx >> kp.spread_shift) * kp.spread_stride;
                            tile = tile >= ntiles ? tile - ntiles : tile;
                        }
                        // the queue walks the 8x8 tiles down the columns, the columns from the
                        // centre outwards (as the one-pixel-per-lane kernels dispatch theirs)
                        const int col = tile / tile_rows8, ty = tile - col * tile_rows8;
                        const int mid8 = tiles_x8 >> 1;
                        const int tx = (col & 1) ? (mid8 - 1 - (col >> 1)) : (mid8 + (col >> 1));
                        px = tx * 8 + (idx & 7);
                        packed_row = ty * 8 + (idx >> 3);
                        if (PILOT) { px *= 2; packed_row *= 2; nrays_pixel = 0; }       // the cell's top-left pixel
                        const int row = (kp.part + (packed_row >> 4) * kp.nparts) * 16 + (packed_row & 15);
                        if (px < kp.w && row < kp.h) {
                            gy = kp.bottom_up ? row : (kp.h - 1 - row);
                            if (MULTI) { accr = accg = accb = 0.0f; sample = 0; si = 0; sj = 0; }
                            start_sample();
                            active = true;
                        }
                    }
                }
                if (cross) { pool_tile = new_tile; pool_next = new_tile >= 0 ? pool_next + nneed - 64 : 64; }
                else pool_next += nneed;
            }
            if (__ballot(active) == 0ull) {
                if (exhausted && pool_next >= 64) break;
                continue;
            }
        }
        // One trace + shading step of the lanes that have a pixel.  Idle lanes skip forward to
        // the end of the loop body (never `continue`: a back edge taken by idle lanes alone
        // would let them spin in a loop of their own while the busy lanes wait to reconverge).
#ifdef RT_DEV_SWITCHES
        dev_trips++;
#endif
        // A wave that is down to its last few rays (deep pixels, nothing left to refill with) puts
        // all of its lanes on one ray at a time instead of walking the hierarchy with most lanes idle.
        constexpr bool COOP = DEEP;
        constexpr bool FLAT = HIER && REFILL && !DEBUG && POOL && RT_FLAT;
        float t_coop = 0.0f; int id_coop = -1;
        bool coop = false;      // this trip's trace is done here, by the whole wave (wave-uniform)
        cnt.tick(HS_T_HEAD);
        if (COOP) {
            const int nrays = __popcll(__ballot(active));
            if (nrays <= kp.coop_rays) {
                trace_cooperative<DEBUG>(sc, L, kp.polish, active, ox, oy, oz, dx, dy, dz, t_coop, id_coop, cnt);
                coop = true;
            }
        }
        cnt.tick(HS_T_COOP);
        if (FLAT && !coop) {
            trace_hierarchy_flat<DEBUG, TOPPOOL, MULTI>(sc, hier, L, kp.polish, flat_lds, active, ox, oy, oz, dx, dy, dz, t_coop, id_coop, cnt);
            coop = true;
#ifdef RT_DEV_SWITCHES
            if (kp.wave_log && kp.dev_stats && active)
                dev_trace_stats<DEBUG>(sc, hier, L, ox, oy, oz, dx, dy, dz, id_coop >= 0 && id_coop < L.n ? t_coop : 1e30f,
                                       state == ST_SHADOW, light_dist - travelled, cnt);
#endif
        }
        do {
        if (REFILL && !active) break;
        if (SYNC && (!active || parked)) break;
        if (PILOT) nrays_pixel++;
        // [census: trace_call @ ITER]
        WCNT(WP_ITER); LCNT(WP_ITER_LANES);
        float t; int id;
        if (FLAT || (COOP && coop)) { t = t_coop; id = id_coop; }
        else if (HIER) trace_hierarchy<DEBUG>(sc, hier, L, kp.polish, queue, ox, oy, oz, dx, dy, dz, t, id, cnt);
        else trace_closest<DEBUG>(lds, L, kp.polish, ox, oy, oz, dx, dy, dz, t, id, cnt);

        bool spawn = false;     // this lane finished shading a hit: spawn children
        bool pop = false;       // this lane needs its next ray from the pending slots

        if (state == ST_PATH) {
            // [census: path_dispatch @ PATH]
            WCNT(WP_PATH); LCNT(WP_PATH_LANES);
            if (DEBUG) {
                if (sample == 0 && seg == kp.capture_seg) { cap_id = (float)id; cap_t = id >= 0 ? t : 0.0f; cap_depth = (float)depth; }
                seg++;
            }
            if (id < 0) {
                // sky (:570-580); mix(h, z, t) = h + t*(z - h); the blue channel is exactly 1
                // [census: sky @ SKY]
                CNT(EV_SEG_MISS);
                WCNT(WP_SKY);
                float tt = 0.5f + 0.5f * dy;
                float sx = 0.8f + tt * (0.5f - 0.8f);
                float sy = 0.9f + tt * (0.7f - 0.9f);
                fr = fr + sx * thx;
                fg = fg + sy * thy;
                fb = fb + thz;
                pop = true;
            } else {
                // [census: hit_info @ HIT]
                CNT(EV_SEG_HIT);
                WCNT(WP_HIT); LCNT(WP_HIT_LANES);
                // ---- HitInfo of the winner (:93-100, :122-141, :357-372)
                hx = ox + t * dx; hy = oy + t * dy; hz = oz + t * dz;
                float h_ior, h_r0, h_eta_in;
                if (id == L.n) {
                    nx = 0.0f; ny = 1.0f; nz = 0.0f;
                    float s = __builtin_floorf(hx) + __builtin_floorf(hz);
                    float two_fl = 2.0f * __builtin_floorf(s * 0.5f);
                    float col = (s == two_fl) ? 0.9f : 0.1f;
                    mr = mg = mb = col; m_refl = 0.05f; m_transp = 0.0f; h_ior = 1.0f;
                    h_r0 = 0.0f; h_eta_in = 1.0f;         // ((1-1)/(1+1))^2 and 1/1
                } else {
                    const float4 *mat = reinterpret_cast<const float4 *>(sc + L.off_material) + 2 * id;
                    float4 m0 = mat[0], m1 = mat[1];
                    mr = m0.x; mg = m0.y; mb = m0.z; m_refl = m0.w; m_transp = m1.x; h_ior = m1.y;
                    h_r0 = m1.z; h_eta_in = m1.w;
                    float qx, qy, qz;
                    object_centre(sc, id, qx, qy, qz);
                    int kd = reinterpret_cast<const int *>(sc + L.off_kind)[id];
                    if (kd < 0) {
                        // [census: normal_sphere @ NORMAL_SPHERE]
                        WCNT(WP_NORMAL_SPHERE);
                        float vx = hx - qx, vy = hy - qy, vz = hz - qz;
                        float rs = rt_rsq(dot3(vx, vy, vz, vx, vy, vz));
                        nx = vx * rs; ny = vy * rs; nz = vz * rs;
                    } else {
                        // [census: normal_torus @ NORMAL_TORUS]
                        WCNT(WP_NORMAL_TORUS);
                        TorusFrame f = load_torus(sc, L.off_torus, kd);
                        float ocx = ox - qx, ocy = oy - qy, ocz = oz - qz;
                        float rox = dot3(ocx, ocy, ocz, f.ux, f.uy, f.uz);
                        float roy = dot3(ocx, ocy, ocz, f.wx, f.wy, f.wz);
                        float roz = dot3(ocx, ocy, ocz, f.vx, f.vy, f.vz);
                        float rdx = dot3(dx, dy, dz, f.ux, f.uy, f.uz);
                        float rdy = dot3(dx, dy, dz, f.wx, f.wy, f.wz);
                        float rdz = dot3(dx, dy, dz, f.vx, f.vy, f.vz);
                        float lx = rox + t * rdx, ly = roy + t * rdy, lz = roz + t * rdz;
                        float alpha = rt_div(f.R, rt_sqrt(lx * lx + lz * lz));
                        float g = 1.0f - alpha;
                        float ax = lx * g, ay = ly, az = lz * g;
                        float rs = rt_rsq(dot3(ax, ay, az, ax, ay, az));
                        ax = ax * rs; ay = ay * rs; az = az * rs;
                        float wx = (ax * f.ux + ay * f.wx) + az * f.vx;
                        float wy = (ax * f.uy + ay * f.wy) + az * f.vy;
                        float wz = (ax * f.uz + ay * f.wz) + az * f.vz;
                        float rs2 = rt_rsq(dot3(wx, wy, wz, wx, wy, wz));
                        nx = wx * rs2; ny = wy * rs2; nz = wz * rs2;
                    }
                }
                // ---- orientation + Fresnel (:583-597, :524-534); dot(d,n) < 0 in compiled form
                // [census: fresnel @ HIT]
                bool entering = (dz * nz + dy * ny) < -(dx * nx);
                osign = entering ? 1.0f : -1.0f;
                float n1 = entering ? cur_ior : h_ior;
                n2 = entering ? h_ior : 1.0f;
                float onx = entering ? nx : -nx, ony = entering ? ny : -ny, onz = entering ? nz : -nz;
                dni = dot3(dx, dy, dz, onx, ony, onz);
                float cos_t = __builtin_fabsf(dni);
                // r0 = ((n1-n2)/(n1+n2))^2 and eta = n1/n2 (:525-528).  Between air (1.0) and the
                // object's medium both are per-object constants, staged with the material by the
                // host with the same binary32 operations: entering from air n1 = 1, n2 = ior;
                // leaving n1 = ior, n2 = 1 (r0 is the same value both ways: the quotient only
                // changes sign, and eta = ior / 1 = ior).  Only a ray that enters from a medium
                // other than air computes them here.
                float r0 = h_r0;
                eta = entering ? h_eta_in : h_ior;
                if (entering && cur_ior != 1.0f) {
                    // [census: fresnel_div @ FRESNEL_DIV]
                    WCNT(WP_FRESNEL_DIV);
                    r0 = rt_div(n1 - n2, n1 + n2); r0 = r0 * r0;
                    eta = rt_div(n1, n2);
                }
                // [census: fresnel_pow @ FRESNEL_POW]
                float sin_t2 = (eta * eta) * (1.0f - cos_t * cos_t);
                if (sin_t2 > 1.0f) { CNT(EV_FRESNEL_TIR); F = 1.0f; }
                else { CNT(EV_FRESNEL_FULL); WCNT(WP_FRESNEL_POW); F = r0 + (1.0f - r0) * rt_pow(1.0f - cos_t, 5.0f); }
                // [census: local_setup @ HIT]
                local = (1.0f - m_refl) - m_transp;
                pdx = dx; pdy = dy; pdz = dz;
                if (local > 0.0f) {
                    // phong_lighting -> calculate_light_attenuation
                    CNT(EV_LOCAL); WCNT(WP_LOCAL);
                    if (MULTI) { li = 0; lj = 0; lsr = lsg = lsb = 0.0f; }
                    setup_shadow();
                    if (kp.hops > 0) state = ST_SHADOW;             // trace the first hop next
                    else spawn = true;                               // zero hops: unattenuated
                } else {
                    spawn = true;
                }
            }
        } else {
            // ---- one shadow hop came back (:453-480)
            // [census: shadow_hop @ SHADOW]
            WCNT(WP_SHADOW); LCNT(WP_SHADOW_LANES);
            bool lit_done = true;       // attenuation finished -> shade now
            if (id >= 0) {
                CNT(EV_HOP_HIT_TEST);
                float reach = t + travelled;
                if (reach < light_dist) {
                    // [census: hop_filter @ HOP_FILTER]
                    CNT(EV_HOP_FILTER);
                    WCNT(WP_HOP_FILTER);
                    float cr, cg, cb, ctr;
                    if (id == L.n) {
                        // the ground is opaque (:128): whichever checker colour, colour * 0 = +0
                        cr = cg = cb = 0.9f; ctr = 0.0f;
                    } else {
                        const float4 *mat = reinterpret_cast<const float4 *>(sc + L.off_material) + 2 * id;
                        float4 m0 = mat[0];
                        cr = m0.x; cg = m0.y; cb = m0.z; ctr = mat[1].x;
                    }
                    filr = filr * (cr * ctr); filg = filg * (cg * ctr); filb = filb * (cb * ctr);
                    if ((filb * filb + filg * filg) == -(filr * filr)) {
                        filr = filg = filb = 0.0f;               // fully blocked (:468-470)
                    } else {
                        // [census: hop_continue @ HOP_CONTINUE]
                        CNT(EV_HOP_CONTINUE);
                        WCNT(WP_HOP_CONTINUE);
                        travelled = reach;
                        hop++;
                        if (hop < kp.hops) {
                            ox = (ox + t * dx) + stx; oy = (oy + t * dy) + sty; oz = (oz + t * dz) + stz;
                            lit_done = false;                    // keep tracing towards the light
                        }
                    }
                }
            }
            // [census: shadow_end @ SHADOW]
            if (lit_done) { spawn = true; state = ST_PATH; }
        }

        if (spawn) {
            // [census: spawn @ SPAWN]
            WCNT(WP_SPAWN); LCNT(WP_SPAWN_LANES);
            if (local > 0.0f) {
                // phong_lighting (:488-514) with attenuation = fil*, light direction = d (still L)
                float ambr = 0.05f * mr, ambg = 0.05f * mg, ambb = 0.05f * mb;
                float lcr = ambr, lcg = ambg, lcb = ambb;
                // (diffuse + specular) * attenuation of the light sample the shadow hops were cast to
                auto light_term = [&](float &tr, float &tg, float &tb) {
                    // [census: phong @ PHONG]
                    CNT(EV_PHONG_FULL);
                    WCNT(WP_PHONG); LCNT(WP_PHONG_LANES);
                    float Lx = dx, Ly = dy, Lz = dz;
                    float nlz = nz * Lz, nly = ny * Ly, nlx = nx * Lx;
                    float diff = max0((nlz + nly) + nlx);
                    float vx = kp.eye[0] - hx, vy = kp.eye[1] - hy, vz = kp.eye[2] - hz;
                    float rs = rt_rsq(dot3(vx, vy, vz, vx, vy, vz));
                    vx = vx * rs; vy = vy * rs; vz = vz * rs;
                    float two_dn = (((-nlz) + (-nly)) + (-nlx)) * 2.0f;
                    float rx = -Lx - two_dn * nx, ry = -Ly - two_dn * ny, rz = -Lz - two_dn * nz;
                    float sd = max0(dot3(vx, vy, vz, rx, ry, rz));
                    float spec = 0.5f * rt_pow(sd, 32.0f);
                    tr = (diff * mr + spec) * filr;
                    tg = (diff * mg + spec) * filg;
                    tb = (diff * mb + spec) * filb;
                };
                // [census: spawn_light @ SPAWN]
                if (!area_light) {
                    if ((filb * filb + filg * filg) != -(filr * filr)) {
                        float tr, tg, tb;
                        light_term(tr, tg, tb);
                        lcr = ambr + tr; lcg = ambg + tg; lcb = ambb + tb;
                    } else {
                        CNT(EV_PHONG_SHADOWED);
                    }
                } else {
                    // this light sample's term, then the next sample's shadow ray -- or, after the
                    // last one, the mean (with zero hops all samples are handled right here)
                    for (;;) {
                        if ((filb * filb + filg * filg) != -(filr * filr)) {
                            float tr, tg, tb;
                            light_term(tr, tg, tb);
                            lsr = lsr + tr; lsg = lsg + tg; lsb = lsb + tb;
                        } else {
                            CNT(EV_PHONG_SHADOWED);
                        }
                        li++;
                        if (li == kp.light_grid) { li = 0; lj++; }
                        if (lj == kp.light_grid) break;
                        setup_shadow();
                        if (kp.hops > 0) { state = ST_SHADOW; spawn = false; break; }
                    }
                    lcr = ambr + lsr * kp.inv_light; lcg = ambg + lsg * kp.inv_light; lcb = ambb + lsb * kp.inv_light;
                }
                if (spawn) {
                    fr = fr + (lcr * local) * thx;
                    fg = fg + (lcg * local) * thy;
                    fb = fb + (lcb * local) * thz;
                }
            }
        }
        if (spawn) {
            // ---- children (:606-643)
            // [census: children @ SPAWN]
            pop = true;
            if (depth < kp.max_depth - 1) {
                WCNT(WP_CHILDREN);
                float onx = osign * nx, ony = osign * ny, onz = osign * nz;
                if (m_transp > 0.0f) {
                    // [census: refract @ REFR]
                    CNT(EV_REFR_TRY);
                    WCNT(WP_REFR);
                    float k2 = 1.0f - eta * (eta * (1.0f - dni * dni));
                    float rx = 0.0f, ry = 0.0f, rz = 0.0f;
                    if (!(k2 < 0.0f)) {
                        float g = eta * dni + rt_sqrt(k2);
                        rx = eta * pdx - g * onx; ry = eta * pdy - g * ony; rz = eta * pdz - g * onz;
                    }
                    bool valid = -(rz * rz + ry * ry) < rx * rx;
                    if (valid && np < kp.stack_size) {
                        // [census: refr_push @ REFR_PUSH]
                        CNT(EV_REFR_PUSH);
                        WCNT(WP_REFR_PUSH);
                        float omf = 1.0f - F;
                        float qx = hx + rx * 0.001f, qy = hy + ry * 0.001f, qz = hz + rz * 0.001f;
                        float ux = ((thx * omf) * m_transp) * mr;
                        float uy = ((thy * omf) * m_transp) * mg;
                        float uz = ((thz * omf) * m_transp) * mb;
                        const int slot = np;
                        if (PEND == PEND_HYBRID && slot >= RT_HYBRID_LDS_SLOTS) {
                            float *q = pend_glb + slot * (11 * 256);
                            q[0 * 256] = qx; q[1 * 256] = qy; q[2 * 256] = qz;
                            q[3 * 256] = rx; q[4 * 256] = ry; q[5 * 256] = rz;
                            q[6 * 256] = ux; q[7 * 256] = uy; q[8 * 256] = uz;
                            q[9 * 256] = n2; q[10 * 256] = __int_as_float(depth + 1);
                        } else if (MEMP) {
                            float *q = pend_mem + slot * (11 * WG);
                            q[0 * WG] = qx; q[1 * WG] = qy; q[2 * WG] = qz;
                            q[3 * WG] = rx; q[4 * WG] = ry; q[5 * WG] = rz;
                            q[6 * WG] = ux; q[7 * WG] = uy; q[8 * WG] = uz;
                            q[9 * WG] = n2; q[10 * WG] = __int_as_float(depth + 1);
                        } else {
#pragma unroll
                            for (int k = 0; k < (MEMP ? 1 : CAP); ++k) {
                                SEL_STORE(pend.ox, qx); SEL_STORE(pend.oy, qy); SEL_STORE(pend.oz, qz);
                                SEL_STORE(pend.dx, rx); SEL_STORE(pend.dy, ry); SEL_STORE(pend.dz, rz);
                                SEL_STORE(pend.tx, ux); SEL_STORE(pend.ty, uy); SEL_STORE(pend.tz, uz);
                                SEL_STORE(pend.ior, n2); SEL_STORE(pend.depth, depth + 1);
                            }
                        }
                        np++;
                    } else if (valid) CNT(EV_DROP);
                }
                // [census: reflect_eval @ CHILDREN]
                CNT(EV_REFL_EVAL);
                float Rt = m_refl + (1.0f - m_refl) * F;
                if (Rt > 0.0f) {
                    if (np < kp.stack_size) {
                        // the reflection ray is the next one visited: continue with it directly
                        // [census: reflect @ REFL]
                        CNT(EV_REFL_PUSH);
                        WCNT(WP_REFL);
                        float two_dni = dni * 2.0f;
                        ox = hx + onx * 0.001f; oy = hy + ony * 0.001f; oz = hz + onz * 0.001f;
                        dx = pdx - two_dni * onx; dy = pdy - two_dni * ony; dz = pdz - two_dni * onz;
                        thx = thx * Rt; thy = thy * Rt; thz = thz * Rt;
                        depth = depth + 1;                       // medium (cur_ior) unchanged (:639)
                        pop = false;
                    } else CNT(EV_DROP);
                }
            }
        }
        if (pop) {
            // [census: pop @ POP]
            WCNT(WP_POP);
            if (np > 0) {
                // [census: pop_load @ POP_LOAD]
                WCNT(WP_POP_LOAD);
                np--;
                const int slot = np;
                if (PEND == PEND_HYBRID && slot >= RT_HYBRID_LDS_SLOTS) {
                    const float *q = pend_glb + slot * (11 * 256);
                    ox = q[0 * 256]; oy = q[1 * 256]; oz = q[2 * 256];
                    dx = q[3 * 256]; dy = q[4 * 256]; dz = q[5 * 256];
                    thx = q[6 * 256]; thy = q[7 * 256]; thz = q[8 * 256];
                    cur_ior = q[9 * 256]; depth = __float_as_int(q[10 * 256]);
                } else if (MEMP) {
                    const float *q = pend_mem + slot * (11 * WG);
                    ox = q[0 * WG]; oy = q[1 * WG]; oz = q[2 * WG];
                    dx = q[3 * WG]; dy = q[4 * WG]; dz = q[5 * WG];
                    thx = q[6 * WG]; thy = q[7 * WG]; thz = q[8 * WG];
                    cur_ior = q[9 * WG]; depth = __float_as_int(q[10 * WG]);
                } else {
#pragma unroll
                    for (int k = 0; k < (MEMP ? 1 : CAP); ++k) {
                        SEL_LOAD(ox, pend.ox); SEL_LOAD(oy, pend.oy); SEL_LOAD(oz, pend.oz);
                        SEL_LOAD(dx, pend.dx); SEL_LOAD(dy, pend.dy); SEL_LOAD(dz, pend.dz);
                        SEL_LOAD(thx, pend.tx); SEL_LOAD(thy, pend.ty); SEL_LOAD(thz, pend.tz);
                        SEL_LOAD(cur_ior, pend.ior); SEL_LOAD(depth, pend.depth);
                    }
                }
            } else {
                // path finished: next stratified sample of this pixel, or done
                // [census: finish @ FINISH]
                WCNT(WP_FINISH);
                if (DEBUG && sample == 0) cap_nseg = seg;
                if (MULTI) {
                    accr = accr + fr; accg = accg + fg; accb = accb + fb;
                    sample++;
                    si++;
                    if (si == N) { si = 0; sj++; }
                    if (sample < nsamples) { if (SYNC) parked = true; else start_sample(); }
                    else active = false;
                } else {
                    active = false;
                }
                if (PILOT) {
                    if (!active) kp.pilot_cost[(size_t)(packed_row >> 1) * kp.cells_x + (px >> 1)] = (unsigned char)(nrays_pixel < 255 ? nrays_pixel : 255);
                } else if (REFILL && !active) write_pixel();
#ifdef RT_DEV_SWITCHES
                if (REFILL && !active) dev_pixels++;
#endif
            }
        }
        } while (false);
    }

#ifdef RT_DEV_SWITCHES
    if (REFILL && kp.wave_log) {
        unsigned long long px_done = dev_pixels;
        for (int o = 32; o > 0; o >>= 1) px_done += __shfl_xor(px_done, o);
        if (lane == 0) {
            unsigned long long *w = kp.wave_log + ((size_t)blockIdx.x * 4 + wave) * RT_WAVE_LOG_WORDS;
            w[0] = dev_t0; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = dev_trips; w[3] = px_done;
        }
        if constexpr (!DEBUG) {
            unsigned long long *w = kp.wave_log + ((size_t)blockIdx.x * 4 + wave) * RT_WAVE_LOG_WORDS;
#ifndef RT_DEV_TIMERS_ONLY
            for (int k = 0; k < HS_T_HEAD; ++k) {
                unsigned long long a = cnt.hs_wave[k], b = cnt.hs_lane[k];
                for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (lane == 0) { w[4 + k] = a; w[4 + HS_COUNT + k] = b; }
            }
#endif
            for (int k = HS_T_HEAD; k < HS_COUNT; ++k)
                if (lane == 0) w[4 + k] = cnt.hs_time[k - HS_T_HEAD];
        }
    }
#endif
    // ---- framebuffer write
    // [census: epilogue @ 1]
    if (!DEBUG) {
        if (!REFILL && in_frame) write_pixel();
    } else {
        if (in_frame && cap_out) {
            size_t o = ((size_t)packed_row * kp.w + px) * 4;
            cap_out[o] = cap_id; cap_out[o + 1] = cap_t; cap_out[o + 2] = cap_depth; cap_out[o + 3] = (float)cap_nseg;
        }
        __syncthreads();
        if (ev_out && threadIdx.x < RT_COUNTER_SLOTS) atomicAdd(&ev_out[threadIdx.x], (unsigned long long)cnt.lds_ev[threadIdx.x]);
    }
}

__global__ void reset_counters_kernel(unsigned int *counters) { counters[threadIdx.x * RT_SUBQUEUE_STRIDE] = 0u; }

// ------------------------------------------------------------------ launchers
template <int CAP, bool DEBUG, int PEND, bool MULTI, bool HIER, bool REFILL, bool DEEP = false, bool POOL = true, bool TOPPOOL = false, bool PILOT = false, bool WHOLE = false>
static int launch_variant(const KParams &kp, const float *d_scene, void *d_out, unsigned long long *d_ev,
                          float *d_cap, hipStream_t stream, LaunchInfo *info)
{
    constexpr bool LDSP = PEND == PEND_LDS || PEND == PEND_HYBRID;
    static_assert(!(LDSP && HIER), "the LDS pending slots and the candidate queue share one region");
    static_assert((PEND != PEND_GLOBAL && PEND != PEND_HYBRID) || REFILL,
                  "the global pending-ray workspace is sliced by persistent workgroup");
    if ((PEND == PEND_GLOBAL || PEND == PEND_HYBRID) && !kp.pend_ws) return (int)hipErrorInvalidValue;
    static_assert(!REFILL || !DEBUG, "the instrumented kernel keeps one pixel per lane");
    static_assert(!DEEP || (HIER && REFILL && !DEBUG), "the deep-frame code belongs to the persistent hierarchy kernels");
    static_assert(POOL || !DEEP, "deep frames always pool");
    static_assert(!TOPPOOL || (POOL && HIER && REFILL && !DEBUG), "the pooled top-node loop is part of the pooled trace");
    static_assert(!PILOT || (HIER && REFILL && POOL && !DEEP && !MULTI && !DEBUG), "the pilot is a shallow single-sample launch of the pooled hierarchy kernel");
    static_assert(!WHOLE || (HIER && REFILL && !TOPPOOL && !PILOT && !DEBUG), "the whole scene in LDS is a choice of the persistent hierarchy kernels for mid-sized scenes");
    auto kfn = rt_kernel<CAP, DEBUG, PEND, MULTI, HIER, REFILL, DEEP, POOL, TOPPOOL, PILOT, WHOLE>;
    size_t lds = (size_t)(kp.scene.total_dwords - ((HIER && !WHOLE) ? kp.scene.off_top : 0)) * 4 + (DEBUG ? RT_COUNTER_SLOTS * 4 : 0);
    constexpr int WG = rt_wg_size<PEND, DEBUG, REFILL>();
    if (LDSP) lds = ((lds + 15) & ~(size_t)15) + (size_t)(PEND == PEND_HYBRID ? RT_HYBRID_LDS_SLOTS : CAP) * 11 * WG * 4;
    if (HIER) lds = ((lds + 15) & ~(size_t)15) + ((REFILL && !DEBUG && POOL && RT_FLAT) ? (size_t)4 * RT_FLAT_DWORDS(TOPPOOL) * 4 : (size_t)RT_QUEUE * 256 * 2);
    // at most three workgroups per CU where the host asks for it: more than a quarter of the CU's LDS each
    if (HIER && kp.three_workgroups_per_cu && lds < 41 * 1024) lds = 41 * 1024;
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    // the attribute is kept per device: set it when this variant first runs on a device
    // (launches of one variant may come from several threads, one per device)
    static std::atomic<unsigned long long> attr_devices{0ull};       // bit d: set on device d
    int cur_device = -1;
    if (hipGetDevice(&cur_device) != hipSuccess || cur_device < 0 || cur_device > 63) return (int)hipErrorInvalidDevice;
    if (!((attr_devices.load(std::memory_order_acquire) >> cur_device) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_devices.fetch_or(1ull << cur_device, std::memory_order_release);
    }
    int bands = (kp.h + 15) / 16;
    int my_bands = (bands - kp.part + kp.nparts - 1) / kp.nparts;
    if (my_bands <= 0 || kp.w <= 0) return 0;
    dim3 grid(my_bands, (kp.w + 15) / 16);       // x: band of this part, y: column slot (see the kernel)
    if (WG == 64) grid = dim3(my_bands * 2, (kp.w + 7) / 8);     // x: 8-row half band, y: 8-pixel column slot
    if (REFILL) {
        // persistent waves: no more workgroups than the queue has work for or the GPU can hold
        if (!kp.work_counter) return (int)hipErrorInvalidValue;
        const int ntiles = PILOT ? ((kp.cells_x + 7) / 8) * my_bands : ((kp.w + 7) / 8) * my_bands * 2;
        int wgs = (ntiles + 3) / 4;
        if (wgs > kp.resident_workgroups) wgs = kp.resident_workgroups;
        grid = dim3(wgs < 1 ? 1 : wgs, 1);
        hipLaunchKernelGGL(reset_counters_kernel, dim3(1), dim3(RT_SUBQUEUES), 0, stream, kp.work_counter);
    }
    hipLaunchKernelGGL(kfn, grid, dim3(WG), lds, stream, kp, d_scene, d_out, d_ev, d_cap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (info) {
        // (asked once per variant: the query costs the host as much as a launch)
        static std::atomic<int> vgprs{-1};
        if (vgprs.load(std::memory_order_relaxed) < 0) {
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn)) == hipSuccess) vgprs.store(fa.numRegs, std::memory_order_relaxed);
        }
        info->vgprs = vgprs.load(std::memory_order_relaxed) < 0 ? 0 : vgprs.load(std::memory_order_relaxed);
        info->lds_bytes = (int)lds;
        info->pending_cap = CAP;
    }
    return 0;
}

#ifndef RT_SCENE_IN_LDS
#define RT_SCENE_IN_LDS 1       // 0: the hierarchy kernels never stage more than the hierarchy (A/B builds)
#endif
// development switch (set by the host API in -DRT_DEV_SWITCHES builds only): 1 = keep the
// pending rays of the one-pixel-per-lane kernels in registers instead of LDS (A/B runs)
static bool dev_no_ldsp = false;
void set_dev_switch(int which, int value) { if (which == 0) dev_no_ldsp = value != 0; }

int launch_render(const KParams &kp, const float *d_scene, void *d_out, void *stream, LaunchInfo *info)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const RtVariantFamily family = rt_variant_family(kp);
    const bool no_ldsp = dev_no_ldsp;
    const bool multi = kp.spp > 1 || kp.light_grid > 1;      // sample loops: pixel samples, light samples
    // deep frames (the host's choice, see make_kparams): pixels spread over waves + cooperative trace
    const bool deep = kp.deep != 0;
    const bool pool = kp.pooled_trace != 0;         // (the host's choice as well)
    const bool toppool = (pool || deep) && kp.pooled_top != 0;
    // Mid-sized scenes: the whole blob in LDS when the workgroup then still takes no more than its share
    // of the CU's LDS (a quarter; a third where the host holds the occupancy at three workgroups).
    // Same-call A/B, everything in LDS against hierarchy only: 96 objects 4K depth 8 1.53 -> 1.45 ms, 128
    // objects depth 4 1.09 -> 1.02 ms (per-lane lists); 64 tori 4K 1.31 -> 1.27 ms (pooled).  At the price
    // of a workgroup per CU it loses (256 spheres 4K depth 16: 260 -> 271 ms; 284 -> 271 at equal
    // occupancy), and with the pooled lists halved to make room it gains nothing there.
    const size_t work_bytes = pool ? (size_t)4 * RT_FLAT_DWORDS(false) * 4 : (size_t)RT_QUEUE * 256 * 2;
    const size_t whole_bytes = (((size_t)kp.scene.total_dwords * 4 + 15) & ~(size_t)15) + work_bytes;
    const bool whole = RT_SCENE_IN_LDS && !toppool && whole_bytes <= (size_t)(kp.three_workgroups_per_cu ? 53 : 40) * 1024;
#define RT_LAUNCH(CAP_, LDSP_)                                                                              \
    (multi ? launch_variant<CAP_, false, (LDSP_) ? PEND_LDS : PEND_REG, true, false, false>(kp, d_scene, d_out, nullptr, nullptr, s, info) \
           : launch_variant<CAP_, false, (LDSP_) ? PEND_LDS : PEND_REG, false, false, false>(kp, d_scene, d_out, nullptr, nullptr, s, info))
#define RT_LAUNCH_HIER_V(CAP_, PEND_, MULTI_, DEEP_, POOL_, TOP_, WHOLE_) \
    launch_variant<CAP_, false, PEND_, MULTI_, true, true, DEEP_, POOL_, TOP_, false, WHOLE_>(kp, d_scene, d_out, nullptr, nullptr, s, info)
#define RT_LAUNCH_HIER_W(CAP_, PEND_, MULTI_, DEEP_, POOL_) \
    (whole ? RT_LAUNCH_HIER_V(CAP_, PEND_, MULTI_, DEEP_, POOL_, false, true) : RT_LAUNCH_HIER_V(CAP_, PEND_, MULTI_, DEEP_, POOL_, false, false))
#define RT_LAUNCH_HIER_M(CAP_, PEND_, MULTI_)                                                              \
    (deep ? (toppool ? RT_LAUNCH_HIER_V(CAP_, PEND_, MULTI_, true, true, true, false) : RT_LAUNCH_HIER_W(CAP_, PEND_, MULTI_, true, true)) \
     : pool ? (toppool ? RT_LAUNCH_HIER_V(CAP_, PEND_, MULTI_, false, true, true, false) : RT_LAUNCH_HIER_W(CAP_, PEND_, MULTI_, false, true)) \
            : RT_LAUNCH_HIER_W(CAP_, PEND_, MULTI_, false, false))
#define RT_LAUNCH_HIER(CAP_, PEND_) (multi ? RT_LAUNCH_HIER_M(CAP_, PEND_, true) : RT_LAUNCH_HIER_M(CAP_, PEND_, false))
    // Large scenes run as persistent waves with pixel refill (measured 1.1-2.4x there: 64 tori
    // 4K 2.43 -> 2.30 ms, 1024 objects 4K depth 4 13.4 -> 10.6 ms, 256 spheres 4K depth 16
    // 2250 -> 928 ms).  The 5-object scene keeps one pixel per lane: there the lanes' loss of
    // lock-step costs more than the idle lanes (depth 8 / stack 8: 0.96 -> 1.53 ms).
    if (family == RT_FAMILY_HIERARCHY) {
        // pending rays in the global workspace: 123 VGPRs whatever the stack size (register slots:
        // 147 / 168 / 256 for 3 / 7 / 16), measured 1.06-1.46x faster than register slots
        return RT_LAUNCH_HIER(RT_PEND_MAX, PEND_GLOBAL);
    }
    if (family == RT_FAMILY_WHOLE_TILES_GLOBAL) {
        // per-object loop forced on a scene whose blob leaves no LDS for pending slots
        // (RT_FLAG_NO_ACCEL with ~1000 tori): all slots in the workspace
        KParams kw = kp;
        kw.whole_tiles = 1;
        return multi ? launch_variant<RT_PEND_MAX, false, PEND_GLOBAL, true, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info)
                     : launch_variant<RT_PEND_MAX, false, PEND_GLOBAL, false, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info);
    }
    if (family == RT_FAMILY_WHOLE_TILES_HYBRID) {
        // Small scene, more than three pending rays possible: persistent waves that take whole
        // tiles (lanes stay in lock-step), slots 0-2 in LDS and the deeper ones in the global
        // workspace.  Keeps the 34 KB / 4-waves-per-SIMD footprint of the three-slot kernel
        // (default scene 4K: depth 8 stack 8 0.60 ms, against 0.91 with seven LDS slots at two
        // workgroups per CU and 0.96 with seven register slots; depth 16 stack 16 2.8 ms
        // against 8.9 with sixteen register slots).
        KParams kw = kp;
        kw.whole_tiles = 1;
        return multi ? launch_variant<RT_PEND_MAX, false, PEND_HYBRID, true, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info)
                     : launch_variant<RT_PEND_MAX, false, PEND_HYBRID, false, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info);
    }
    // 3 slots x 11 dwords x 256 threads = 33 KB of LDS per workgroup: fits 4 workgroups per CU
    // next to a small scene, and frees the registers for a fourth wave per SIMD
    if (!no_ldsp) return RT_LAUNCH(3, true);
    return RT_LAUNCH(3, false);
#undef RT_LAUNCH
#undef RT_LAUNCH_HIER
#undef RT_LAUNCH_HIER_M
#undef RT_LAUNCH_HIER_V
#undef RT_LAUNCH_HIER_W
}

// The pilot launch of a deep frame (see KParams::pilot_cost): kp carries the small depth and the cost buffer.
int launch_pilot(const KParams &kp, const float *d_scene, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!kp.pilot_cost || kp.cells_x <= 0) return (int)hipErrorInvalidValue;
    if (kp.pooled_top)
        return launch_variant<RT_PEND_MAX, false, PEND_GLOBAL, false, true, true, false, true, true, true>(kp, d_scene, nullptr, nullptr, nullptr, s, nullptr);
    return launch_variant<RT_PEND_MAX, false, PEND_GLOBAL, false, true, true, false, true, false, true>(kp, d_scene, nullptr, nullptr, nullptr, s, nullptr);
}

int launch_debug(const KParams &kp, const float *d_scene, unsigned long long *d_events, float *d_capture,
                 void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return launch_variant<16, true, PEND_REG, true, false, false>(kp, d_scene, nullptr, d_events, d_capture, s, nullptr);
}

#ifndef RT_FAST
// Exhaustive comparison of the range-restricted sqrt / reciprocal / division cores with the
// compiler's full expansions.  counts[0]: sqrt over all x in [2^-96, 2^96); counts[1]:
// 1/sqrt over the same x; counts[2]: 1/s over all s in [2^-48, 2^48]; counts[3]: log2's
// (m - 1)/(m + 1) over all m in [1, 2).  A count is the number of operands whose results
// differ in any bit.  counts[4], counts[5]: x / 3 and x / 27 over all x of both signs with |x|
// in [2^-96, 2^96).
__global__ __launch_bounds__(256) void rt_selftest_kernel(unsigned long long *counts)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t stride = gridDim.x * 256u;
    unsigned int bad0 = 0, bad1 = 0, bad2 = 0, bad3 = 0;
    for (uint32_t u = 0x0f800000u + tid; u < 0x6f800000u; u += stride) {
        float x = bits_f(u);
        bad0 += f_bits(sqrt_core(x)) != f_bits(__builtin_sqrtf(x));
        bad1 += f_bits(rcp_core(sqrt_core(x))) != f_bits(1.0f / __builtin_sqrtf(x));
        if (u - stride < 0x0f800000u) {      // first trip of this thread: the guards themselves
            bad0 += !sqrt_in_core_range(x);
            bad0 += sqrt_in_core_range(bits_f(0x0f7fffffu)) + sqrt_in_core_range(bits_f(0x6f800000u)) +
                    sqrt_in_core_range(0.0f) + sqrt_in_core_range(-1.0f) + sqrt_in_core_range(bits_f(0x7f800000u)) +
                    sqrt_in_core_range(bits_f(0x7fc00000u)) + sqrt_in_core_range(bits_f(0x00000001u));
        }
    }
    for (uint32_t u = 0x27800000u + tid; u <= 0x57800000u; u += stride) {     // 2^-48 .. 2^48
        float sv = bits_f(u);
        bad2 += f_bits(rcp_core(sv)) != f_bits(1.0f / sv);
    }
    for (uint32_t u = 0x3f800000u + tid; u < 0x40000000u; u += stride) {
        float m = bits_f(u);
        bad3 += f_bits(div_core(m - 1.0f, m + 1.0f)) != f_bits((m - 1.0f) / (m + 1.0f));
    }
    unsigned int bad4 = 0, bad5 = 0;
    for (uint32_t u = 0x0f800000u + tid; u < 0x6f800000u; u += stride) {
        for (uint32_t sign = 0; sign < 2; ++sign) {
            float x = bits_f(u | (sign << 31));
            bad4 += f_bits(div_const_core<3>(x)) != f_bits(x / 3.0f);
            bad5 += f_bits(div_const_core<27>(x)) != f_bits(x / 27.0f);
            bad4 += !div_const_in_core_range(x);
        }
    }
    if (tid == 0)
        bad4 += div_const_in_core_range(0.0f) + div_const_in_core_range(-0.0f) + div_const_in_core_range(bits_f(0x0f7fffffu)) +
                div_const_in_core_range(bits_f(0xef800000u)) + div_const_in_core_range(bits_f(0x7f800000u)) +
                div_const_in_core_range(bits_f(0xffc00000u)) + div_const_in_core_range(bits_f(0x80000001u));
    if (bad4) atomicAdd(&counts[4], (unsigned long long)bad4);
    if (bad5) atomicAdd(&counts[5], (unsigned long long)bad5);
    if (bad0) atomicAdd(&counts[0], (unsigned long long)bad0);
    if (bad1) atomicAdd(&counts[1], (unsigned long long)bad1);
    if (bad2) atomicAdd(&counts[2], (unsigned long long)bad2);
    if (bad3) atomicAdd(&counts[3], (unsigned long long)bad3);
}

int launch_selftest(unsigned long long *d_counts, void *stream)
{
    hipLaunchKernelGGL(rt_selftest_kernel, dim3(256 * 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_counts);
    return (int)hipGetLastError();
}
#endif

}  // namespace RT_NS
