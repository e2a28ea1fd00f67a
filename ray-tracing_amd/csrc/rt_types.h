// Internal types shared by the host API (rt_api.hip) and the device code (rt_device.hip).
#pragma once
#include <stdint.h>

// ---- staged scene ("blob"): what the host packs, what every workgroup copies to LDS ----
//
// Layout in dwords, all sections 16-byte aligned:
//   [0, 8*np)            bound[j]    = { cx0 cx1 cy0 cy1 } { cz0 cz1 br0 br1 }: objects 2j and 2j+1
//                                      interleaved so that the bounding test of both runs as packed
//                                      FP32 math; br = bounding_radius^2; np = ceil(n/2); an odd n
//                                      is padded with a record whose br is -inf (never passes)
//   [.., + 8n)           material[i] = { r, g, b, reflectivity } { transparency, ior, r0, 1/ior }: r0 = ((1-ior)/(1+ior))^2
//   [.., + 16*nt)        torus[s]    = { ux uy uz wx } { wy wz vx vy } { vz R R^2 r^2 } { 4R^2 8R^2 r 0 }
//   [.., + n)            kind[i]     = -1 for a sphere, torus slot s >= 0 for a torus
// Large scenes add a two-level hierarchy of bounding spheres over the objects' bounding
// spheres (objects sorted into spatial order; a leaf node holds RT_LEAF consecutive slots, a
// top node RT_FAN consecutive leaf nodes):
//   [.., + 8*ceil(n2/2)) top[j]      = { cx0 cx1 cy0 cy1 } { cz0 cz1 R0 R1 }: nodes 2j, 2j+1
//   [.., + 8*ceil(n1/2)) leaf[j]     = same record, leaf nodes 2j, 2j+1
//   [.., + 8*(n1+1)*RT_LEAF/2) member[s] = bound records (as bound[]) of the objects in slot order;
//                                      empty slots carry br = -inf; the slots of leaf n1 are all empty
//   [.., + (n1+1)*RT_LEAF/2) slot_id[s] = object index of slot s, 16 bits each; bit 15: the object is a torus
// Keeping only the 16 bytes per object of the bound record in the per-object hot loop and
// everything else behind the bound test is what makes 1024 objects fit in LDS next to
// several workgroups.
struct SceneLayout {
    int n, ntori;
    int off_material;   // dword offsets
    int off_torus;
    int off_kind;
    int n1, n2;         // leaf / top nodes of the sphere hierarchy (0: none)
    int off_top, off_leaf, off_member, off_slot_id;
    float hc[3], hR;    // sphere that contains every object's bounding sphere and every node
    float hq[3];        // the per-object loop's scene-sphere test as a polynomial in |o - hc|^2 (see scene_test_polynomial)
    int total_dwords;   // padded to a multiple of 4
};

// The scene-sphere test of the per-object loop (trace_closest) asks whether a ray of near-unit direction d
// from o can pass any object's binary32 bounding test: with q = o - hc it keeps the loop unless
//   (q.d)^2 + T(|q|^2) - |q|^2 < 0,   T = (hR + sf u)^2 + 2.02 sf2 u^2 + 1e-30,   u = 1.0001 (|q|^2 / 2 + 1/2 + hR)
// (u bounds the distance |q| + hR to the far side of the sphere without a square root; sf, sf2: the slack
// of the node tests for directions with | |d|^2 - 1 | <= 1e-5).  T is a quadratic polynomial in |q|^2 with
// positive coefficients; they are computed here in double precision and rounded up with a relative margin
// of 1e-5, far above the two roundings of its evaluation with fused multiply-adds, so the kernel's value
// is never below T.
static inline void scene_test_polynomial(SceneLayout &L)
{
    const double sf = 3.75e-3, sf2 = 1.42e-5, hR = (double)L.hR;
    const double g = sf * sf + 2.02 * sf2, k0 = 1.0001 * (0.5 + hR), k1 = 1.0001 * 0.5;
    const double c[3] = {hR * hR + 2.0 * hR * sf * k0 + g * k0 * k0 + 1e-30, 2.0 * hR * sf * k1 + 2.0 * g * k0 * k1, g * k1 * k1};
    for (int i = 0; i < 3; ++i) L.hq[i] = (float)(c[i] * (1.0 + 1e-5)) * 1.000001f;
}

#define RT_LEAF 8        // object slots per leaf node (even)
#define RT_FAN 8         // leaf nodes per top node (even)
#define RT_SUBQUEUES 32        // interleaved sub-queues of the persistent-wave work queue, one counter each
#define RT_SUBQUEUE_STRIDE 64   // dwords between two counters (each on its own 256-byte line)
#define RT_PEND_MAX 16   // pending-ray slots per lane in the global workspace (= RT_MAX_PENDING)
#ifndef RT_MAX_LIGHT_GRID
#define RT_MAX_LIGHT_GRID 8   // area light: at most 8 x 8 samples (as in rt355.h)
#endif
#define RT_QUEUE 32      // per-ray candidate queue entries (16 bits each, in LDS)

static inline SceneLayout scene_layout(int n, int ntori, bool hierarchy)
{
    SceneLayout L;
    L.n = n; L.ntori = ntori;
    int np = (n + 1) / 2;
    L.off_material = 8 * np;
    L.off_torus = L.off_material + 8 * n;
    L.off_kind = L.off_torus + 16 * ntori;
    int end = (L.off_kind + n + 3) & ~3;
    L.n1 = L.n2 = 0;
    L.hc[0] = L.hc[1] = L.hc[2] = 0.0f; L.hR = 0.0f;
    L.hq[0] = L.hq[1] = L.hq[2] = 0.0f;
    L.off_top = L.off_leaf = L.off_member = L.off_slot_id = end;
    if (hierarchy) {
        L.n1 = (n + RT_LEAF - 1) / RT_LEAF;
        L.n2 = (L.n1 + RT_FAN - 1) / RT_FAN;
        L.off_top = end;
        L.off_leaf = L.off_top + 8 * ((L.n2 + 1) / 2);
        L.off_member = L.off_leaf + 8 * ((L.n1 + 1) / 2);
        // one leaf more than there are leaf nodes: empty slots only (what a lane without work tests)
        L.off_slot_id = L.off_member + 8 * ((L.n1 + 1) * RT_LEAF / 2);
        end = (L.off_slot_id + (L.n1 + 1) * RT_LEAF / 2 + 3) & ~3;
    }
    L.total_dwords = end;
    return L;
}

// ---- kernel arguments (passed by value: lands in the kernarg segment / SGPRs) ----
struct KParams {
    float right[3], up[3], fwd[3], eye[3];   // columns of u_camera_to_world (fragment_shader.glsl:656-659)
    float focal;
    float light[3];
    float res_w, res_h;
    float inv_res_h;        // RN(1 / res_h)
    int uv_by_reciprocal;   // the host has checked, for every pixel centre of this frame size, that the short
                            // reciprocal sequence returns the correctly rounded quotient by res_h
    int w, h;
    int max_depth, stack_size, hops, spp;
    int bottom_up;          // output row order
    int part, nparts;       // band split
    int my_bands;           // 16-row bands this part renders
    unsigned int *work_counter;   // queue heads of the persistent-wave kernels: RT_SUBQUEUES counters,
                                  // RT_SUBQUEUE_STRIDE dwords apart, zeroed ahead of every launch
    int resident_workgroups;      // workgroups the GPU can hold at once (persistent-wave grid size)
    int whole_tiles;              // persistent waves take whole 8x8 tiles only (no per-lane refill)
    float *pend_ws;               // pending-ray workspace: resident_workgroups x pend_ws_slots x 11 x 256 floats
    int pend_ws_slots;            // slots per lane in that workspace (RT_PEND_MAX, or fewer in the static scratch)
    int out_mode;           // 0: RGBA8, 1: float4, 2: RGB8 (3 bytes per pixel, packed)
    int polish;             // Newton polish on torus roots
    int capture_seg;        // debug flavour only
    int wave_profile;       // debug flavour only: count executed code blocks per wave instead of events
    float sample_off[16];   // (i + 0.5)/spp - 0.5 for i < spp, rounded to binary32
    int light_grid;         // G x G samples of the square area light; 1: the reference's point light
    float inv_light;        // 1 / G^2
    float light_off[RT_MAX_LIGHT_GRID];   // ((i + 0.5)/G - 0.5) * light_size for i < G
    SceneLayout scene;
    int use_hierarchy;      // walk the sphere hierarchy instead of the per-object loop
    int spread_shift;       // persistent kernels: log2 of the lanes that take neighbouring pixels (6: a wave's 64 pixels
    int spread_stride;      // are one 8x8 tile); smaller: groups from tiles spread_stride apart in the queue order
    int three_workgroups_per_cu;   // hierarchy kernels: hold the occupancy at 3 waves per SIMD (small deep frames: the
                            // frame waits for its longest waves, which run faster with fewer neighbours)
    int pooled_top;         // with pooled_trace: pool the loop over the kept top nodes as well (more than 64 leaf nodes)
    int pooled_trace;       // hierarchy kernels: pool the wave's member tests and intersections (trace_hierarchy_flat)
    int coop_rays;          // persistent hierarchy kernels: a wave with at most this many rays left traces them
                            // one at a time with all lanes (0: never)
    int deep;               // deep frame: run the instantiation that carries the spread / cooperative trace / pixel order
    // Deep frames, cost-ordered pixels.  A pilot launch of the same kernel (PILOT instantiation) traces
    // the top-left pixel of every 2x2 pixel cell of this part to a small depth and stores the number of
    // rays it took (pilot_cost, one byte per cell, cells in packed-row order, cells_x per row); two
    // small kernels turn that into `order`: the cells sorted by that number, largest first, each as
    // (cell row << 16) | cell column.  The render launch then hands out pixels in that order -- queue
    // entry e = the 16 cells order[16 e ..] -- so the pixels with the deepest ray trees start first and
    // the frame does not end with a few lanes walking through them while the rest of the GPU idles.
    unsigned char *pilot_cost;      // pilot launch: where the ray counts go
    const unsigned int *order;      // render launch: the sorted cells (null: tile order)
    int order_cells;                // cells in `order`
    int order_deal;                 // queue entries per block of the dealt order (the persistent waves of the launch)
    int cells_x;                    // cells per row of this part: ceil(w / 2)
#ifdef RT_DEV_SWITCHES
    int dev_stats;                  // development builds: collect the cut-off statistics of the pooled trace (slow)
    unsigned long long *wave_log;   // development builds: per persistent wave {start, end (100 MHz ticks), loop trips, pixels,
                                    // hierarchy-walk loop iterations per wave [6] and per lane [6]}, RT_WAVE_LOG_WORDS words
#endif
};

// ---- work events (same classes, same weights as the shader-source FLOP convention) ----
enum RtEvent {
    EV_PIXEL_PATH, EV_TRACE, EV_BOUND_CULL_DISC, EV_BOUND_TBOUND, EV_BOUND_TORUS_RADIUS,
    EV_SPHERE_CALL, EV_SPHERE_ROOT, EV_SPHERE_HIT,
    EV_PLANE_CALL, EV_PLANE_T, EV_PLANE_HIT,
    EV_TORUS_CALL, EV_QUAD_FULL, EV_QUAD_EARLY, EV_BIQ_Z, EV_ROOT_OUT,
    EV_CUBIC, EV_CARDANO, EV_TRIG, EV_FERRARI_TAIL, EV_TORUS_HIT,
    EV_SEG_MISS, EV_SEG_HIT, EV_FRESNEL_TIR, EV_FRESNEL_FULL, EV_LOCAL,
    EV_PHONG_SHADOWED, EV_PHONG_FULL, EV_ATTEN_FIXED, EV_HOP_HIT_TEST, EV_HOP_FILTER, EV_HOP_CONTINUE,
    EV_REFR_TRY, EV_REFR_PUSH, EV_REFL_EVAL, EV_REFL_PUSH, EV_DROP,
    EV_COUNT
};
#define RT_WAVE_LOG_WORDS 64
#define RT_COUNTER_SLOTS 64      // counters the instrumented kernel keeps (events, or the wave profile's blocks)
static_assert(EV_COUNT <= RT_COUNTER_SLOTS, "event counters");

// Blocks of the render loop counted by the wave profile (development builds, -DRT_DEV_SWITCHES):
// NAME counts the waves that enter the block with at least one lane, NAME_LANES the lanes.
#define RT_WAVE_PROFILE_BLOCKS(X)                                                                    \
    X(ITER) X(ITER_LANES) X(PAIR) X(CAND) X(CAND_LANES) X(TBOUND) X(SPHERE) X(SPHERE_LANES) X(SPHERE_SOLVE)   \
    X(SPHERE_ROOT2) X(TORUS) X(TORUS_LANES) X(BIQUADRATIC) X(CUBIC) X(CARDANO) X(CARDANO_LANES) X(TRIG) X(TRIG_LANES) \
    X(SCENE_SKIP) X(FERRARI_TAIL) X(QUAD_OFFER) X(QUAD_ROOTS) X(PLANE_DIV) X(PATH) X(PATH_LANES) X(SKY) X(HIT) X(HIT_LANES)  \
    X(NORMAL_SPHERE) X(NORMAL_TORUS) X(FRESNEL_POW) X(FRESNEL_DIV) X(LOCAL) X(SHADOW) X(SHADOW_LANES)   \
    X(HOP_FILTER) X(HOP_CONTINUE) X(SPAWN) X(SPAWN_LANES) X(PHONG) X(PHONG_LANES) X(CHILDREN) X(REFR)     \
    X(REFR_PUSH) X(REFL) X(POP) X(POP_LOAD) X(FINISH) X(START)
enum RtWaveProfileBlock {
#define X(n) WP_##n,
    RT_WAVE_PROFILE_BLOCKS(X)
#undef X
    WP_COUNT
};
static_assert(WP_COUNT <= RT_COUNTER_SLOTS, "wave profile counters");

// Launchers, one set per numerics flavour (rt_device.hip is compiled twice).
struct LaunchInfo { int vgprs; int lds_bytes; int pending_cap; };

// Which family of kernel variants renders a frame with these parameters -- one rule, used by the
// launchers (to pick the variant) and by the entry points (the persistent-wave variants share the
// pending-ray workspace and the tile-queue counters, so their launches are ordered on the GPU).
enum RtVariantFamily {
    RT_FAMILY_ONE_PIXEL_PER_LANE,     // small scene, at most three pending rays: slots in LDS
    RT_FAMILY_WHOLE_TILES_HYBRID,     // small scene, deeper stacks: persistent waves, slots in LDS + workspace
    RT_FAMILY_WHOLE_TILES_GLOBAL,     // per-object loop on a scene whose blob leaves no LDS for slots
    RT_FAMILY_HIERARCHY               // large scene: sphere hierarchy, persistent waves with per-lane refill
};
inline int rt_needed_pending(const KParams &kp)
{
    const int need = kp.stack_size < kp.max_depth - 1 ? kp.stack_size : kp.max_depth - 1;
    return need < 1 ? 1 : need;
}
inline RtVariantFamily rt_variant_family(const KParams &kp)
{
    if (kp.use_hierarchy && kp.scene.n1 > 0) return RT_FAMILY_HIERARCHY;
    // 3 slots x 11 dwords x 256 threads = 33 KB of LDS per workgroup next to the scene blob
    if ((size_t)kp.scene.total_dwords * 4 + 3 * 11 * 256 * 4 > 150 * 1024) return RT_FAMILY_WHOLE_TILES_GLOBAL;
    if (rt_needed_pending(kp) > 3) return RT_FAMILY_WHOLE_TILES_HYBRID;
    return RT_FAMILY_ONE_PIXEL_PER_LANE;
}
inline bool rt_uses_workspace(const KParams &kp) { return rt_variant_family(kp) != RT_FAMILY_ONE_PIXEL_PER_LANE; }

#define RT_DECLARE_FLAVOUR(NS)                                                                   \
    namespace NS {                                                                               \
    int launch_render(const KParams &kp, const float *d_scene, void *d_out, void *stream,       \
                      LaunchInfo *info);                                                         \
    int launch_pilot(const KParams &kp, const float *d_scene, void *stream);                    \
    int launch_debug(const KParams &kp, const float *d_scene, unsigned long long *d_events,     \
                     float *d_capture, void *stream);                                            \
    void set_dev_switch(int which, int value);                                                   \
    }

namespace rt_strict { int launch_selftest(unsigned long long *d_counts, void *stream); }
RT_DECLARE_FLAVOUR(rt_strict)
RT_DECLARE_FLAVOUR(rt_fast)
