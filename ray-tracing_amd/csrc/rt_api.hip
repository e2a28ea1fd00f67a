// rt_api.hip -- host side of librt355.so: the C ABI declared in include/rt355.h.
//
// Takes over what App.render does between the scene list and the back buffer
// (main.py:277-303): parameter validation, per-object invariants, scene staging on the
// GPU, kernel launch, read-back.  Host arithmetic that feeds the kernels is plain binary32
// with no contraction (this file is built with -ffp-contract=off) so that the derived
// per-object values are the ones the reference shader computes per ray.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/rt355.h"
#include "rt_types.h"

// Scratch of rt_render_device_noalloc: the tile-queue counters and a small pending-ray workspace
// in one file-scope device array (every GPU the library is loaded on holds its own copy for the
// life of the process).  Host code never passes this name to a kernel: rt_init looks the
// device address up once per device (hipGetSymbolAddress) and everything uses that pointer.
#define RT_NOALLOC_COUNTER_BYTES (RT_SUBQUEUES * RT_SUBQUEUE_STRIDE * sizeof(unsigned int))
#define RT_NOALLOC_WS_BYTES ((size_t)RT_NOALLOC_WORKGROUPS * RT_NOALLOC_MAX_PENDING * 11 * 256 * sizeof(float))
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_COUNTER_BYTES + RT_NOALLOC_WS_BYTES];

namespace {

struct State {
    bool inited = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_pending = false;
    float *d_scene = nullptr;
    size_t d_scene_cap = 0;
    SceneLayout layout{};
    bool have_scene = false;
    void *d_out = nullptr;          // staging for rt_render / debug launches
    size_t d_out_cap = 0;
    unsigned long long *d_events = nullptr;
    unsigned int *d_work = nullptr;     // tile queue head of the persistent-wave kernels
    int resident_workgroups = 2048;
    float *d_pend_ws = nullptr;         // pending-ray workspace of the persistent-wave kernels
    hipEvent_t ws_event = nullptr;      // end of the last launch that used the workspace
    bool ws_busy = false;
    // deep frames: ray counts of the pilot launch (one byte per 2x2 pixel cell), the cells sorted by
    // them, and the 2 x 256 counters of the sort (see KParams::pilot_cost)
    unsigned char *d_pilot_cost = nullptr;
    unsigned int *d_order = nullptr, *d_order_counters = nullptr;
    size_t order_cap = 0;               // cells the three buffers hold
#ifdef RT_DEV_SWITCHES
    unsigned long long *d_wave_log = nullptr;   // RT_WAVE_LOG_WORDS words per resident wave (development builds)
#endif
    std::vector<rt_object> scene_objs;      // the objects of the staged scene (rt_set_scene skips an unchanged scene)
    // rt_init_multi: one context per part of the frame.  Part i is rendered on parts[i].device into
    // its own packed buffer and copied from there, band by band, into the frame on the first device.
    struct Part {
        int device = -1;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;          // this part's bands are in the frame
        hipEvent_t frame_free = nullptr;    // recorded on the frame's stream: the frame may be overwritten
        hipEvent_t t_start = nullptr, t_kernel = nullptr, t_copy = nullptr;     // a timed frame: before the kernel, after it, after the copy
        bool timed = false;                 // the three events of the last frame were recorded
        float *d_scene = nullptr; size_t d_scene_cap = 0;
        void *d_part = nullptr; size_t d_part_cap = 0;
        unsigned int *d_work = nullptr;
        float *d_pend_ws = nullptr;
        int resident_workgroups = 0;
    };
    std::vector<Part> parts;                // empty: single-device mode
    unsigned char *scratch_ptr = nullptr;   // device address of `scratch` on g.device
    hipEvent_t scratch_event = nullptr;     // end of the last launch that used the static scratch
    bool scratch_busy = false;
    rt_timing timing{};
    char err[512] = "";
} g;

// Parts 1.. are issued by one thread each (bound to the part's device once), part 0 by the caller:
// issuing a part costs the host about 0.04 ms (kernel launch, strided copy, events), which in
// sequence would take longer than the frame's kernels run once there are eight parts.
struct Issuers {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable wake, finished;
    unsigned long long generation = 0;      // bumped per frame
    int pending = 0;                        // parts of this frame not issued yet
    bool stop = false;
    const rt_params *params = nullptr;      // the frame being issued
    void *d_frame = nullptr;
    bool timed = false;                     // record the parts' timing events
    std::vector<int> rc;                    // per part: result of the last issue
    std::vector<std::string> err;           //           and its message
} g_issuers;


// where fail() writes: the library's message buffer, or the issuing thread's own one
thread_local char *t_err = nullptr;

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err ? t_err : g.err, sizeof g.err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail(RT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// dot/normalize in the reference's operation order (z, y, then x; v * (1/sqrt(dot)))
inline float dot3(const float *a, const float *b) { return (a[2] * b[2] + a[1] * b[1]) + a[0] * b[0]; }
inline void normalize3(float *v)
{
    float r = 1.0f / std::sqrt(dot3(v, v));
    v[0] *= r; v[1] *= r; v[2] *= r;
}
inline void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

int ensure_out(size_t bytes)
{
    if (bytes <= g.d_out_cap) return 0;
    if (g.d_out) HIP_TRY(hipFree(g.d_out));
    g.d_out = nullptr; g.d_out_cap = 0;
    HIP_TRY(hipMalloc(&g.d_out, bytes));
    g.d_out_cap = bytes;
    return 0;
}

int validate_params(const rt_params *p)
{
    if (!p) return fail(RT_E_ARG, "params is null");
    if (p->width <= 0 || p->height <= 0 || p->width > 32768 || p->height > 32768)
        return fail(RT_E_ARG, "resolution %dx%d out of range", p->width, p->height);
    if (p->max_depth < 1) return fail(RT_E_ARG, "max_depth must be >= 1");
    // a path tree can double with every level while the stack has room: bound the depth so that
    // one call cannot occupy the GPU for minutes
    if (p->max_depth > RT_MAX_DEPTH) return fail(RT_E_LIMIT, "max_depth %d exceeds the limit of %d", p->max_depth, RT_MAX_DEPTH);
    if (p->stack_size < 1) return fail(RT_E_ARG, "stack_size must be >= 1");
    if (p->shadow_hops < 0 || p->shadow_hops > 64) return fail(RT_E_ARG, "shadow_hops out of range");
    if (p->spp_grid < 1 || p->spp_grid > 16) return fail(RT_E_LIMIT, "spp_grid must be 1..16");
    int need = p->stack_size < p->max_depth - 1 ? p->stack_size : p->max_depth - 1;
    if (need > RT_MAX_PENDING)
        return fail(RT_E_LIMIT, "min(stack_size, max_depth-1) = %d exceeds the %d pending-ray slots per lane",
                    need, RT_MAX_PENDING);
    if (p->nparts < 1 || p->part < 0 || p->part >= p->nparts) return fail(RT_E_ARG, "bad part/nparts");
    if (p->light_grid < 0 || p->light_grid > RT_MAX_LIGHT_GRID)
        return fail(RT_E_LIMIT, "light_grid must be 0..%d", RT_MAX_LIGHT_GRID);
    if (!(p->light_size >= 0.0f) || !std::isfinite(p->light_size)) return fail(RT_E_ARG, "light_size must be finite and >= 0");
    if ((p->flags & RT_FLAG_OUT_F32) && (p->flags & RT_FLAG_OUT_RGB8))
        return fail(RT_E_ARG, "RT_FLAG_OUT_F32 and RT_FLAG_OUT_RGB8 exclude each other");
    for (int i = 0; i < 16; ++i)
        if (!std::isfinite(p->cam_to_world[i])) return fail(RT_E_ARG, "camera matrix is not finite");
    if (!std::isfinite(p->focal_length)) return fail(RT_E_ARG, "focal length is not finite");
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(p->light_pos[i])) return fail(RT_E_ARG, "light position is not finite");
    return 0;
}

int part_rows(const rt_params *p)
{
    int bands = (p->height + RT_BAND_ROWS - 1) / RT_BAND_ROWS;
    int rows = 0;
    for (int b = p->part; b < bands; b += p->nparts) {
        int r0 = b * RT_BAND_ROWS;
        int r1 = r0 + RT_BAND_ROWS < p->height ? r0 + RT_BAND_ROWS : p->height;
        rows += r1 - r0;
    }
    return rows;
}

// The primary ray's uv = (2 (x + 0.5) - w, 2 (y + 0.5) - h) / h (fragment_shader.glsl:650-653) has a
// divisor that is the same for the whole frame.  q = n * RN(1/h), corrected once by the exact
// remainder, is the correctly rounded n / h for almost all operands; this runs that sequence (the
// kernel's div_by_reciprocal, same operations in binary32 with fused multiply-adds) over every
// numerator the frame's pixel centres produce and says whether all of them came out equal to n / h.
// The kernel takes the short sequence only then.  Checked once per frame size.
bool uv_quotients_by_reciprocal(int w, int h)
{
    static std::mutex mu;
    static int cw = 0, ch = 0;
    static bool cok = false;
    std::lock_guard<std::mutex> lk(mu);
    if (w == cw && h == ch) return cok;
    const float fw = (float)w, fh = (float)h, rd = 1.0f / fh;
    bool ok = true;
    auto same = [&](float n) {
        const float q = n * rd;
        const float r = std::fmaf(-fh, q, n);
        const float v = std::fmaf(r, rd, q), want = n / fh;
        return std::memcmp(&v, &want, 4) == 0;
    };
    for (int x = 0; x < w && ok; ++x) ok = same(((float)x + 0.5f) * 2.0f - fw);
    for (int y = 0; y < h && ok; ++y) ok = same(((float)y + 0.5f) * 2.0f - fh);
    cw = w; ch = h; cok = ok;
    return ok;
}

int coop_rays_for(int n);
KParams make_kparams(const rt_params *p)
{
    KParams k{};
    const float *m = p->cam_to_world;
    // columns of the matrix as the shader sees it after the transposed upload
    for (int r = 0; r < 3; ++r) {
        k.right[r] = m[4 * r + 0];
        k.up[r] = m[4 * r + 1];
        k.fwd[r] = m[4 * r + 2];
        k.eye[r] = m[4 * r + 3];
    }
    k.focal = p->focal_length;
    for (int i = 0; i < 3; ++i) k.light[i] = p->light_pos[i];
    k.res_w = (float)p->width; k.res_h = (float)p->height;
    k.inv_res_h = 1.0f / k.res_h;
    k.uv_by_reciprocal = uv_quotients_by_reciprocal(p->width, p->height) ? 1 : 0;
    k.w = p->width; k.h = p->height;
    k.max_depth = p->max_depth; k.stack_size = p->stack_size; k.hops = p->shadow_hops; k.spp = p->spp_grid;
    k.bottom_up = (p->flags & RT_FLAG_BOTTOM_UP) ? 1 : 0;
    k.part = p->part; k.nparts = p->nparts;
    {
        int bands = (p->height + RT_BAND_ROWS - 1) / RT_BAND_ROWS;
        k.my_bands = (bands - p->part + p->nparts - 1) / p->nparts;
    }
    k.work_counter = g.d_work;      // launches that use the queue are ordered on the GPU (workspace event)
    k.resident_workgroups = g.resident_workgroups;
    k.pend_ws = g.d_pend_ws;
    k.pend_ws_slots = RT_PEND_MAX;
#ifdef RT_DEV_SWITCHES
    k.wave_log = g.d_wave_log;
#endif
    k.light_grid = p->light_grid > 1 ? p->light_grid : 1;
    k.inv_light = 1.0f / (float)(k.light_grid * k.light_grid);
    for (int i = 0; i < RT_MAX_LIGHT_GRID; ++i)
        k.light_off[i] = i < k.light_grid ? (float)(((double)i + 0.5) / (double)k.light_grid - 0.5) * p->light_size : 0.0f;
    k.out_mode = (p->flags & RT_FLAG_OUT_F32) ? 1 : (p->flags & RT_FLAG_OUT_RGB8) ? 2 : 0;
    k.polish = (p->flags & RT_FLAG_NEWTON_POLISH) ? 1 : 0;
    k.capture_seg = -1;
    k.wave_profile = 0;
    for (int i = 0; i < 16; ++i)
        k.sample_off[i] = i < p->spp_grid ? (float)(((double)i + 0.5) / (double)p->spp_grid - 0.5) : 0.0f;
    k.scene = g.layout;
    k.use_hierarchy = (g.layout.n1 > 0 && !(p->flags & RT_FLAG_NO_ACCEL)) ? 1 : 0;
    // Deep path trees (the frame time is then set by the waves that hold the deepest pixels, not by
    // throughput): hand out the pixels of 64 different tiles to a wave, and let a wave that is down to
    // a few rays trace them one at a time with all lanes.  Shallow frames keep whole 8x8 tiles per
    // wave: there the coherence of neighbouring primary rays is worth more (64 tori 4K depth 8:
    // 2.2 ms with tiles, 3.3 ms spread; 256 spheres 1080p depth 16: 319 -> 172 ms spread).
    const bool deep = p->max_depth >= 12 && std::min(p->stack_size, p->max_depth - 1) >= 8 && k.use_hierarchy;
    // The pooled trace has a fixed cost per trace (ray table, prefix sums, per-ray keys) that a cheap
    // walk does not earn back.  Same-call A/B against per-lane lists (4K unless noted): 48 objects
    // depth 4 +17 %, 96 objects depth 4 / 8 +11 / +5 %; 256 objects depth 8 -10 %, 256 spheres 1080p
    // depth 8 -23 %, 1024 objects depth 8 -20 % (long, uneven leaf lists); 64 tori depth 4 / 8
    // -8 / -17 % (few leaves, but the torus intersections of a wave fill 1.4 rounds instead of 2.8
    // iterations with 21 lanes).
    k.pooled_trace = (deep || g.layout.n >= 192 || g.layout.ntori >= 48) ? 1 : 0;
    k.pooled_top = (k.pooled_trace && g.layout.n1 > 64) ? 1 : 0;     // measured: 1024 objects -4 to -6 %, 512 objects +11 %
    k.deep = deep ? 1 : 0;
    k.cells_x = (p->width + 1) / 2;
    k.spread_shift = deep ? 0 : 6;
    // measured with four waves per SIMD: 256 spheres depth 16 at 1080p 136 -> 146 ms, at 4K 351 -> 332 ms
    k.three_workgroups_per_cu = (deep && (long long)p->width * p->height < 4000000ll) ? 1 : 0;
    k.coop_rays = deep ? coop_rays_for(g.layout.n) : 0;
#ifdef RT_DEV_SWITCHES
    k.dev_stats = getenv("RT_DEV_STATS") != nullptr;
    if (const char *e = getenv("RT_SPREAD_SHIFT")) k.spread_shift = atoi(e);
    if (const char *e = getenv("RT_COOP_RAYS")) k.coop_rays = atoi(e);
#endif
    k.spread_stride = 0;
    if (k.spread_shift < 0 || k.spread_shift > 6) k.spread_shift = 6;
    if (k.spread_shift < 6) {
        const int ntiles = ((p->width + 7) / 8) * k.my_bands * 2;
        const int groups = 64 >> k.spread_shift;
        k.spread_stride = ntiles / groups;      // (groups - 1) * stride < ntiles: one conditional subtraction in the kernel
        if (k.spread_stride < 1) k.spread_shift = 6;
    }
    return k;
}

// Scenes with at least this many objects get the sphere hierarchy; below it the packed
// per-object loop is cheaper than a traversal.
int hierarchy_min_objects()
{
#ifdef RT_DEV_SWITCHES
    // development builds: RT_HIERARCHY_MIN overrides the threshold (experiments)
    static const int v = [] {
        const char *e = getenv("RT_HIERARCHY_MIN");
        return e ? atoi(e) : 40;
    }();
    return v;
#else
    return 40;
#endif
}

// Rays a wave of the persistent hierarchy kernels may have left when it starts tracing them one at a
// time with all 64 lanes (trace_cooperative: n / 64 rounds of object tests per ray, against one
// pooled hierarchy trace for all of the wave's rays together).  Measured: 256 objects best at 2-4
// (1080p depth 16 -10 % against never); 1024 objects, 16 rounds per ray, lose with any threshold
// (+7 % at 1, +23 % at 3).
int coop_rays_for(int n)
{
    return n <= 320 ? 3 : 0;
}

// the hierarchy walk keeps its lists of kept nodes in bit masks: 128 leaf nodes, 32 top nodes
static_assert(RT_MAX_OBJECTS <= 128 * RT_LEAF && RT_MAX_OBJECTS <= 32 * RT_LEAF * RT_FAN && RT_FAN == 8,
              "node masks of trace_hierarchy are sized for this");
static_assert(RT_MAX_OBJECTS <= 32768, "slot ids keep the object index in 15 bits (bit 15 flags a torus)");
static_assert(RT_MAX_OBJECTS / RT_LEAF + 1 <= 256, "the pooled trace's (ray, leaf) items hold the leaf number, or that of the empty leaf behind the last one, in 8 bits");

// Effective bounding radius of an object: the square root of the binary32 value br*br the
// reference's bounding test uses (fragment_shader.glsl:391-402).
double bound_radius(const rt_object &o)
{
    float br = o.type == 1 ? o.radius : o.major_radius + o.minor_radius;
    float br2 = br * br;
    return std::sqrt((double)br2);
}

// Spatial order for the hierarchy: recursive median split along the longest axis of the
// centres' bounding box; split points fall on multiples of RT_LEAF (of RT_LEAF*RT_FAN higher
// up) so that consecutive leaf nodes, and consecutive groups of RT_FAN of them, are subtrees.
void spatial_order(const rt_object *objs, int *idx, int lo, int hi)
{
    int cnt = hi - lo;
    if (cnt <= RT_LEAF) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = lo; k < hi; ++k)
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], objs[idx[k]].center[a]);
            mx[a] = std::max(mx[a], objs[idx[k]].center[a]);
        }
    int axis = 0;
    if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
    if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
    int unit = cnt > RT_LEAF * RT_FAN ? RT_LEAF * RT_FAN : RT_LEAF;
    int half = ((cnt / 2 + unit - 1) / unit) * unit;
    if (half >= cnt) half = cnt - unit > 0 ? cnt - unit : cnt / 2;
    int mid = lo + half;
    std::nth_element(idx + lo, idx + mid, idx + hi, [&](int x, int y) {
        float cx = objs[x].center[axis], cy = objs[y].center[axis];
        return cx < cy || (cx == cy && x < y);
    });
    spatial_order(objs, idx, lo, mid);
    spatial_order(objs, idx, mid, hi);
}

// Bounding sphere {cx, cy, cz, R} of the bounding spheres of objects idx[lo..hi).  The centre
// starts at the centre of their axis-aligned box and is walked towards the farthest sphere with
// shrinking steps (Badoiu-Clarkson), keeping the best centre seen: typically 10-25 % less
// radius than the box centre gives, i.e. fewer nodes kept per ray.  R = max(|c_i - centre| +
// br_i) for the float32 centre actually stored, rounded up.
void node_sphere(const rt_object *objs, const int *idx, int lo, int hi, float out[4])
{
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int k = lo; k < hi; ++k) {
        const rt_object &o = objs[idx[k]];
        double br = bound_radius(o);
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], (double)o.center[a] - br);
            mx[a] = std::max(mx[a], (double)o.center[a] + br);
        }
    }
    auto radius_at = [&](const float c[3], int *farthest) {
        double R = 0.0;
        for (int k = lo; k < hi; ++k) {
            const rt_object &o = objs[idx[k]];
            double dx = (double)o.center[0] - c[0], dy = (double)o.center[1] - c[1], dz = (double)o.center[2] - c[2];
            double e = std::sqrt(dx * dx + dy * dy + dz * dz) + bound_radius(o);
            if (e > R) { R = e; if (farthest) *farthest = k; }
        }
        return R;
    };
    float best[3];
    for (int a = 0; a < 3; ++a) best[a] = (float)(0.5 * (mn[a] + mx[a]));
    double bestR = radius_at(best, nullptr);
    double c[3] = {best[0], best[1], best[2]};
    for (int it = 1; it <= 64 && hi - lo > 1; ++it) {
        float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
        int far = lo;
        double R = radius_at(cf, &far);
        if (R < bestR && std::isfinite(R)) { bestR = R; best[0] = cf[0]; best[1] = cf[1]; best[2] = cf[2]; }
        // step towards the point of the farthest sphere that is farthest from the centre
        const rt_object &o = objs[idx[far]];
        double d[3] = {(double)o.center[0] - c[0], (double)o.center[1] - c[1], (double)o.center[2] - c[2]};
        double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (!(len > 0.0)) break;
        double reach = len + bound_radius(o);
        for (int a = 0; a < 3; ++a) c[a] += d[a] / len * reach / (it + 1.0);
    }
    float Rf = (float)(bestR * (1.0 + 1e-6));
    if ((double)Rf < bestR) Rf = std::nextafter(Rf, INFINITY);
    out[0] = best[0]; out[1] = best[1]; out[2] = best[2]; out[3] = Rf;
}

// Fills the hierarchy sections of the blob.
void build_hierarchy(const rt_object *objs, int n, SceneLayout &L, float *blob)
{
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    spatial_order(objs, idx.data(), 0, n);
    // A node whose members are all spheres stores its radius negated (the sign is a flag: such a
    // node may be skipped when it lies entirely behind the ray, see trace_hierarchy).
    auto put_node = [&](int off, int j, const float s[4], int lo, int hi) {
        bool spheres_only = true;
        for (int k = lo; k < hi; ++k) spheres_only = spheres_only && objs[idx[k]].type == 1;
        float *r = blob + off + 8 * (j >> 1) + (j & 1);
        r[0] = s[0]; r[2] = s[1]; r[4] = s[2]; r[6] = spheres_only ? -s[3] : s[3];
    };
    float s[4], all[4];
    node_sphere(objs, idx.data(), 0, n, all);
    double reach = all[3];
    auto cover = [&](const float nd[4]) {    // the scene sphere must contain the node spheres too
        double dx = (double)nd[0] - all[0], dy = (double)nd[1] - all[1], dz = (double)nd[2] - all[2];
        reach = std::max(reach, (std::sqrt(dx * dx + dy * dy + dz * dz) + nd[3]) * (1.0 + 1e-6));
    };
    for (int j = 0; j < L.n2; ++j) {
        int lo = j * RT_LEAF * RT_FAN, hi = std::min(n, lo + RT_LEAF * RT_FAN);
        node_sphere(objs, idx.data(), lo, hi, s);
        put_node(L.off_top, j, s, lo, hi);
        cover(s);
    }
    for (int j = 0; j < L.n1; ++j) {
        int lo = j * RT_LEAF, hi = std::min(n, lo + RT_LEAF);
        node_sphere(objs, idx.data(), lo, hi, s);
        put_node(L.off_leaf, j, s, lo, hi);
        cover(s);
    }
    L.hc[0] = all[0]; L.hc[1] = all[1]; L.hc[2] = all[2];
    L.hR = std::nextafter((float)reach, INFINITY);
    scene_test_polynomial(L);
    uint16_t *slot_id = reinterpret_cast<uint16_t *>(blob + L.off_slot_id);
    for (int sl = 0; sl < (L.n1 + 1) * RT_LEAF; ++sl) {
        float *r = blob + L.off_member + 8 * (sl >> 1) + (sl & 1);
        if (sl < n) {
            const float *src = blob + 8 * (idx[sl] >> 1) + (idx[sl] & 1);   // the object's own bound record
            r[0] = src[0]; r[2] = src[2]; r[4] = src[4]; r[6] = src[6];
            // bit 15: the object is a torus (indices are below 1024)
            slot_id[sl] = (uint16_t)(idx[sl] | (objs[idx[sl]].type == 1 ? 0 : 0x8000));
        } else {
            r[0] = r[2] = r[4] = 0.0f; r[6] = -INFINITY;
            slot_id[sl] = 0;
        }
    }
}

const char *const k_event_names[EV_COUNT] = {
    "pixel_path", "trace", "bound_cull_disc", "bound_tbound", "bound_torus_radius",
    "sphere_call", "sphere_root", "sphere_hit", "plane_call", "plane_t", "plane_hit",
    "torus_call", "quad_full", "quad_early", "biq_z", "root_out", "cubic", "cardano", "trig",
    "ferrari_tail", "torus_hit", "seg_miss", "seg_hit", "fresnel_tir", "fresnel_full", "local",
    "phong_shadowed", "phong_full", "atten_fixed", "hop_hit_test", "hop_filter", "hop_continue",
    "refr_try", "refr_push", "refl_eval", "refl_push", "drop"};

// Scalar FP32 operations of the reference shader source per event: add/sub/mul/div/sqrt/
// inversesqrt/floor/min/max/mod/pow/cos/acos count 1; compare/negate/abs/sign/select 0;
// dot3 5, cross 9, normalize 9, length 6, reflect 12, refract 22, mix(vec3) 9, clamp 2.
const int k_event_flops[EV_COUNT] = {
    30, 0, 17, 19, 1,        // main() :652-668; bounding test :399-410; torus radius :395
    25, 4, 18,               // intersect_sphere :77-81, :87/:89, :95-96
    5, 2, 10,                // intersect_plane :116, :118, :124/:137
    131, 11, 4, 2, 1,        // intersect_torus set-up + quartic depress :314-339/:252-256; solve_quadratic
    26, 10, 23, 9, 54,       // resolvent + solve_cubic :281/:201-215; Cardano; trig; Ferrari tail; torus hit :359-367
    17, 12, 9, 14, 9,        // sky :575-578; hit common; Fresnel TIR / full :525-533; local accumulate :603
    8, 73, 39, 1, 11, 7,     // phong shadowed / full :490-513; attenuation :443-448, :457, :462-468, :473-474
    28, 16, 3, 21, 0};       // refraction :613-620; reflection :629-637; drops

// ---- deep frames: the order in which the render launch hands out its pixels (see KParams::pilot_cost)
// Counting sort of the cells by the pilot's ray count, largest first: histogram, exclusive scan from the
// top, scatter (inside one count the order is whatever the atomics make of it; the frame does not
// depend on the order of its pixels).
__global__ void order_hist_kernel(const unsigned char *__restrict__ cost, int ncells, unsigned int *__restrict__ hist)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ncells; i += gridDim.x * 256) atomicAdd(&h[cost[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void order_scan_kernel(const unsigned int *__restrict__ hist, unsigned int *__restrict__ cursor)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    unsigned int above = 0u;                                     // cells with a larger count come first
    for (int c = 255; c > (int)threadIdx.x; --c) above += h[c];
    cursor[threadIdx.x] = above;
}
__global__ void order_scatter_kernel(const unsigned char *__restrict__ cost, int ncells, int cells_x,
                                     unsigned int *__restrict__ cursor, unsigned int *__restrict__ order)
{
    __shared__ unsigned int h[256], base[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned int c = 0u, rank = 0u;
    if (i < ncells) { c = cost[i]; rank = atomicAdd(&h[c], 1u); }
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (i < ncells) order[base[c] + rank] = ((unsigned int)(i / cells_x) << 16) | (unsigned int)(i % cells_x);
}

// Depth of the pilot's ray trees.  The pixels whose trees are deep at the frame's depth are those whose
// first few segments keep hitting glass, which this depth already shows.
int pilot_depth()
{
#ifdef RT_DEV_SWITCHES
    if (const char *e = getenv("RT_PILOT_DEPTH")) return atoi(e);
#endif
    return 6;
}

// Pilot launch + sort for a deep frame: fills kp.order / kp.order_cells (buffers d_cost, d_order,
// d_counters of at least `cells` entries / 512 counters) and turns the spread off.  Everything is
// enqueued on `s` ahead of the render launch.
int enqueue_pixel_order(KParams &kp, bool fast, const float *d_scene, unsigned char *d_cost, unsigned int *d_order,
                        unsigned int *d_counters, hipStream_t s)
{
    const int cells_x = kp.cells_x, cells_y = kp.my_bands * (RT_BAND_ROWS / 2);
    const int ncells = cells_x * cells_y;
    KParams pk = kp;
    pk.max_depth = kp.max_depth < pilot_depth() ? kp.max_depth : pilot_depth();
    pk.spp = 1; pk.light_grid = 1; pk.inv_light = 1.0f;
    pk.deep = 0; pk.spread_shift = 6; pk.spread_stride = 0; pk.coop_rays = 0; pk.three_workgroups_per_cu = 0;
    pk.pooled_trace = 1;
    pk.pilot_cost = d_cost;
    HIP_TRY(hipMemsetAsync(d_cost, 0, (size_t)ncells, s));                  // cells the frame's edge cuts off stay at 0
    HIP_TRY(hipMemsetAsync(d_counters, 0, 512 * sizeof(unsigned int), s));
    int lrc = fast ? rt_fast::launch_pilot(pk, d_scene, s) : rt_strict::launch_pilot(pk, d_scene, s);
    if (lrc) return fail(RT_E_HIP, "pilot launch failed: %s", hipGetErrorString((hipError_t)lrc));
    const int blocks = (ncells + 255) / 256;
    hipLaunchKernelGGL(order_hist_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, s, d_cost, ncells, d_counters);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(256), 0, s, d_counters, d_counters + 256);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, s, d_cost, ncells, cells_x, d_counters + 256, d_order);
    HIP_TRY(hipGetLastError());
    kp.order = d_order;
    kp.order_cells = ncells;
    {
        // the render launch's persistent waves: four per workgroup, as many workgroups as launch_variant starts
        int wgs = ((ncells + 15) / 16 + 3) / 4;
        if (wgs > kp.resident_workgroups) wgs = kp.resident_workgroups;
        kp.order_deal = 4 * (wgs < 1 ? 1 : wgs);
#ifdef RT_DEV_SWITCHES
        if (getenv("RT_ORDER_DEAL")) kp.order_deal = atoi(getenv("RT_ORDER_DEAL")) > 0 ? atoi(getenv("RT_ORDER_DEAL")) : 1;
#endif
    }
    kp.spread_shift = 6; kp.spread_stride = 0;
    return RT_OK;
}

// Which deep frames get the pilot and the cost order: those with many pixels per lane of the GPU, whose
// time is the work divided by the lanes that are kept busy (256 spheres 4K depth 16, 32 pixels per lane:
// 281 -> 260 ms).  With few pixels per lane the frame lasts as long as the ray tree of its deepest pixel
// takes one lane, and there the spread does as well or better (1080p, 8 pixels per lane: 256 spheres
// 119 -> 124 ms, 1024 objects depth 16 44 -> 49 ms), so those keep it.
bool wants_pixel_order(const KParams &kp, unsigned int flags)
{
    long long per_lane_x16 = 16ll * kp.resident_workgroups * 256;
    if (flags & RT_FLAG_PIXEL_ORDER) per_lane_x16 = 1;
#ifdef RT_DEV_SWITCHES
    if (getenv("RT_NO_PILOT")) return false;
    if (getenv("RT_PILOT_ALWAYS")) per_lane_x16 = 4096;
#endif
    return kp.deep && kp.use_hierarchy && (long long)kp.w * kp.my_bands * RT_BAND_ROWS >= per_lane_x16;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g.err; }

static void release_parts()
{
    if (!g_issuers.threads.empty()) {
        {
            std::lock_guard<std::mutex> lk(g_issuers.m);
            g_issuers.stop = true;
        }
        g_issuers.wake.notify_all();
        for (auto &t : g_issuers.threads) t.join();
        g_issuers.threads.clear();
        g_issuers.stop = false;
    }
    for (auto &pt : g.parts) {
        if (pt.device < 0) continue;
        (void)hipSetDevice(pt.device);
        if (pt.stream) (void)hipStreamSynchronize(pt.stream);
        if (pt.d_scene) (void)hipFree(pt.d_scene);
        if (pt.d_part) (void)hipFree(pt.d_part);
        if (pt.d_work) (void)hipFree(pt.d_work);
        if (pt.d_pend_ws) (void)hipFree(pt.d_pend_ws);
        if (pt.done) (void)hipEventDestroy(pt.done);
        if (pt.frame_free) (void)hipEventDestroy(pt.frame_free);
        if (pt.t_start) (void)hipEventDestroy(pt.t_start);
        if (pt.t_kernel) (void)hipEventDestroy(pt.t_kernel);
        if (pt.t_copy) (void)hipEventDestroy(pt.t_copy);
        if (pt.stream) (void)hipStreamDestroy(pt.stream);
    }
    g.parts.clear();
    if (g.device >= 0) (void)hipSetDevice(g.device);
}

static void release_all()
{
    release_parts();
    if (g.d_scene) (void)hipFree(g.d_scene);
    if (g.d_out) (void)hipFree(g.d_out);
    if (g.d_events) (void)hipFree(g.d_events);
    if (g.d_work) (void)hipFree(g.d_work);
    if (g.d_pend_ws) (void)hipFree(g.d_pend_ws);
    if (g.d_pilot_cost) (void)hipFree(g.d_pilot_cost);
    if (g.d_order) (void)hipFree(g.d_order);
    if (g.d_order_counters) (void)hipFree(g.d_order_counters);
    if (g.ev0) (void)hipEventDestroy(g.ev0);
    if (g.ev1) (void)hipEventDestroy(g.ev1);
    if (g.ws_event) (void)hipEventDestroy(g.ws_event);
    if (g.scratch_event) (void)hipEventDestroy(g.scratch_event);
    if (g.stream) (void)hipStreamDestroy(g.stream);
    char err[sizeof g.err];
    std::memcpy(err, g.err, sizeof err);
    g = State{};
    std::memcpy(g.err, err, sizeof err);
}

// The pending-ray workspace of the persistent-wave kernels (184 MB on a 256-CU GPU), allocated
// when the first scene or frame that needs it arrives.
static int ensure_workspace()
{
    if (g.d_pend_ws) return 0;
    static_assert(RT_PEND_MAX == RT_MAX_PENDING, "workspace slots per lane");
    size_t ws = (size_t)g.resident_workgroups * RT_PEND_MAX * 11 * 256 * sizeof(float);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_pend_ws), ws));
    return 0;
}

static int ensure_order_buffers(size_t cells)
{
    if (cells <= g.order_cap && g.d_order) return 0;
    if (g.d_pilot_cost) HIP_TRY(hipFree(g.d_pilot_cost));
    if (g.d_order) HIP_TRY(hipFree(g.d_order));
    g.d_pilot_cost = nullptr; g.d_order = nullptr; g.order_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_pilot_cost), cells));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_order), cells * sizeof(unsigned int)));
    if (!g.d_order_counters) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_order_counters), 512 * sizeof(unsigned int)));
    g.order_cap = cells;
    return 0;
}

static int init_device(int device);

int rt_init(int device)
{
    if (g.inited) {
        if (device == g.device) return RT_OK;
        return fail(RT_E_STATE, "already initialised on device %d; call rt_shutdown first (rt_init_multi drives several GPUs)", g.device);
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RT_E_HIP, "no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(RT_E_ARG, "device %d out of range (0..%d)", device, count - 1);
    int rc = init_device(device);
    if (rc) release_all();       // nothing half-made stays behind
    return rc;
}

static int init_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&g.ev0));
    HIP_TRY(hipEventCreate(&g.ev1));
    HIP_TRY(hipEventCreateWithFlags(&g.ws_event, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&g.scratch_event, hipEventDisableTiming));
    {
        // device address of the static scratch array on THIS device (each GPU has its own copy)
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RT_E_HIP, "hipGetSymbolAddress(scratch) returned a null pointer");
        g.scratch_ptr = static_cast<unsigned char *>(ptr);
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_events), RT_COUNTER_SLOTS * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_work), RT_SUBQUEUES * RT_SUBQUEUE_STRIDE * sizeof(unsigned int)));
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        // the persistent-wave kernels run at up to 4 waves per SIMD = 4 workgroups per CU
        g.resident_workgroups = prop.multiProcessorCount * 4;
    }
#ifdef RT_DEV_SWITCHES
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_wave_log), (size_t)g.resident_workgroups * 4 * RT_WAVE_LOG_WORDS * sizeof(unsigned long long)));
    rt_strict::set_dev_switch(0, getenv("RT_NO_LDSP") != nullptr);
    rt_fast::set_dev_switch(0, getenv("RT_NO_LDSP") != nullptr);
#endif
    g.device = device;
    g.inited = true;
    g.err[0] = 0;
    return RT_OK;
}

void rt_shutdown(void)
{
    if (!g.inited) return;
    (void)hipSetDevice(g.device);
    (void)hipDeviceSynchronize();
    release_all();
    g.err[0] = 0;
}

// Validation + staging of a scene on the host: the blob the kernels read and its layout.
static int pack_scene_host(const rt_object *objs, int n, SceneLayout &L_out, std::vector<float> &blob_out)
{
    if (n < 0 || (n > 0 && !objs)) return fail(RT_E_ARG, "bad scene pointer/count");
    if (n > RT_MAX_OBJECTS) return fail(RT_E_SCENE, "%d objects exceed the limit of %d", n, RT_MAX_OBJECTS);
    int ntori = 0;
    for (int i = 0; i < n; ++i) {
        const rt_object &o = objs[i];
        // any other type reads an uninitialised HitInfo in the reference (fragment_shader.glsl:416-421)
        if (o.type != 1 && o.type != 2) return fail(RT_E_SCENE, "object %d: type %d is not 1 (sphere) or 2 (torus)", i, o.type);
        const float *f = &o.center[0];
        for (int k = 0; k < 15; ++k)
            if (!std::isfinite(f[k])) return fail(RT_E_SCENE, "object %d: field %d is not finite", i, k + 1);
        if (o.type == 2) ntori++;
    }
    SceneLayout L = scene_layout(n, ntori, n >= hierarchy_min_objects() && n > RT_LEAF);
    std::vector<float> blob((size_t)L.total_dwords + 4, 0.0f);
    int *kind = reinterpret_cast<int *>(blob.data() + L.off_kind);
    int slot = 0;
    if (n & 1) blob[8 * (n >> 1) + 1 + 6] = -INFINITY;   // pad record of the last pair: disc = -inf
    for (int i = 0; i < n; ++i) {
        const rt_object &o = objs[i];
        // bounding radius: radius, or major + minor for a torus (fragment_shader.glsl:391-396)
        float br = o.type == 1 ? o.radius : o.major_radius + o.minor_radius;
        float *b = blob.data() + 8 * (i >> 1) + (i & 1);
        b[0] = o.center[0]; b[2] = o.center[1]; b[4] = o.center[2]; b[6] = br * br;
        float *m = blob.data() + L.off_material + 8 * i;
        m[0] = o.color[0]; m[1] = o.color[1]; m[2] = o.color[2]; m[3] = o.reflectivity;
        m[4] = o.transparency; m[5] = o.refractive_index;
        // Fresnel constants against air (fragment_shader.glsl:525-528 with n1 = 1, n2 = ior)
        float q = (1.0f - o.refractive_index) / (1.0f + o.refractive_index);
        m[6] = q * q;
        m[7] = 1.0f / o.refractive_index;
        if (o.type == 1) { kind[i] = -1; continue; }
        kind[i] = slot;
        // torus frame (fragment_shader.glsl:315-317): w = normalize(normal),
        // u = normalize(cross(w, |w.y| > 0.99 ? (1,0,0) : (0,1,0))), v = cross(w, u)
        float w[3] = {o.normal[0], o.normal[1], o.normal[2]};
        normalize3(w);
        float sel = std::fabs(w[1]) > 0.99f ? 1.0f : 0.0f;
        float helper[3] = {sel, 1.0f - sel, 0.0f};
        float u[3], v[3];
        cross3(w, helper, u);
        normalize3(u);
        cross3(w, u, v);
        float R = o.major_radius, r = o.minor_radius;
        float *t = blob.data() + L.off_torus + 16 * slot;
        t[0] = u[0]; t[1] = u[1]; t[2] = u[2]; t[3] = w[0];
        t[4] = w[1]; t[5] = w[2]; t[6] = v[0]; t[7] = v[1];
        t[8] = v[2]; t[9] = R; t[10] = R * R; t[11] = r * r;
        t[12] = (4.0f * R) * R; t[13] = (8.0f * R) * R; t[14] = r;
        slot++;
    }
    if (L.n1 > 0) build_hierarchy(objs, n, L, blob.data());
    else if (n > 0) {
        // sphere around all objects' bounding spheres: rays that miss it skip the object loop
        std::vector<int> idx(n);
        for (int i = 0; i < n; ++i) idx[i] = i;
        float all[4];
        node_sphere(objs, idx.data(), 0, n, all);
        L.hc[0] = all[0]; L.hc[1] = all[1]; L.hc[2] = all[2];
        L.hR = std::isfinite(all[3]) ? std::nextafter(all[3], INFINITY) : 0.0f;
        scene_test_polynomial(L);
    }
    L_out = L;
    blob_out.swap(blob);
    return RT_OK;
}

int rt_set_scene(const rt_object *objs, int n)
{
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    // the same objects as the staged scene (a viewer re-submits its scene every frame, as
    // main.py:283-296 re-uploads the uniforms): nothing to do
    if (g.have_scene && n >= 0 && (size_t)n == g.scene_objs.size() &&
        (n == 0 || (objs && std::memcmp(objs, g.scene_objs.data(), (size_t)n * sizeof(rt_object)) == 0)))
        return RT_OK;
    SceneLayout L;
    std::vector<float> blob;
    int rc = pack_scene_host(objs, n, L, blob);
    if (rc) return rc;
    size_t bytes = (size_t)L.total_dwords * sizeof(float);
    if (bytes < 16) bytes = 16;
    if (bytes > g.d_scene_cap) {
        if (g.d_scene) HIP_TRY(hipFree(g.d_scene));
        g.d_scene = nullptr; g.d_scene_cap = 0;
        size_t cap = bytes < 4096 ? 4096 : bytes;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_scene), cap));
        g.d_scene_cap = cap;
    }
    // blocking copy: later launches on any stream see the new scene
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(g.d_scene, blob.data(), bytes, hipMemcpyHostToDevice));
    // from here until every copy is in place there is no current scene: a failure below must not
    // leave a state in which a retry with the same objects is taken for "unchanged"
    g.have_scene = false;
    g.scene_objs.clear();
    g.layout = L;
    // every device that renders a part holds its own copy of the scene
    auto stage_part = [&](State::Part &pt) -> int {
        HIP_TRY(hipSetDevice(pt.device));
        if (bytes > pt.d_scene_cap) {
            if (pt.d_scene) HIP_TRY(hipFree(pt.d_scene));
            pt.d_scene = nullptr; pt.d_scene_cap = 0;
            size_t cap = bytes < 4096 ? 4096 : bytes;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pt.d_scene), cap));
            pt.d_scene_cap = cap;
        }
        HIP_TRY(hipStreamSynchronize(pt.stream));
        HIP_TRY(hipMemcpy(pt.d_scene, blob.data(), bytes, hipMemcpyHostToDevice));
        return RT_OK;
    };
    for (auto &pt : g.parts) {
        rc = stage_part(pt);
        if (rc) { (void)hipSetDevice(g.device); return rc; }
    }
    if (!g.parts.empty()) HIP_TRY(hipSetDevice(g.device));
    if (L.n1 > 0) {                              // large scenes render as persistent waves
        rc = ensure_workspace();
        if (rc) return rc;
    }
    g.have_scene = true;
    g.scene_objs.assign(objs, objs + n);
    return RT_OK;
}

int rt_stage_scene_host(const rt_object *objs, int n, float *blob, size_t blob_dwords, int32_t layout[16])
{
    SceneLayout L;
    std::vector<float> staged;
    int rc = pack_scene_host(objs, n, L, staged);
    if (rc) return rc;
    if (!layout) return fail(RT_E_ARG, "layout is null");
    const int32_t v[16] = {L.n, L.ntori, L.off_material, L.off_torus, L.off_kind, L.total_dwords, L.n1, L.n2,
                           L.off_top, L.off_leaf, L.off_member, L.off_slot_id, RT_LEAF, RT_FAN, 0, 0};
    std::memcpy(layout, v, sizeof v);
    float hs[4] = {L.hc[0], L.hc[1], L.hc[2], L.hR};
    if (blob) {
        if (blob_dwords < (size_t)L.total_dwords + 4) return fail(RT_E_SIZE, "blob needs %d dwords", L.total_dwords + 4);
        std::memcpy(blob, staged.data(), (size_t)L.total_dwords * sizeof(float));
        std::memcpy(blob + L.total_dwords, hs, sizeof hs);
    }
    return RT_OK;
}

size_t rt_output_bytes(const rt_params *p)
{
    if (!p || p->width <= 0 || p->height <= 0 || p->nparts < 1 || p->part < 0 || p->part >= p->nparts) return 0;
    size_t bpp = (p->flags & RT_FLAG_OUT_F32) ? 16 : (p->flags & RT_FLAG_OUT_RGB8) ? 3 : 4;
    return (size_t)part_rows(p) * (size_t)p->width * bpp;
}

static int render_multi(const rt_params *p, void *d_frame, hipStream_t s, bool timed);

int rt_render_device(const rt_params *p, void *d_out, size_t out_bytes, void *stream, int timed)
{
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    if (!g.have_scene) return fail(RT_E_STATE, "no scene set (rt_set_scene)");
    int rc = validate_params(p);
    if (rc) return rc;
    size_t need = rt_output_bytes(p);
    if (!d_out && need) return fail(RT_E_ARG, "output pointer is null");
    if (out_bytes < need) return fail(RT_E_SIZE, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : g.stream;
    if (!g.parts.empty()) return render_multi(p, d_out, s, timed != 0);      // (timing per part: rt_get_part_timing)
    KParams kp = make_kparams(p);
    LaunchInfo info{};
    // the persistent-wave kernels of large scenes share one pending-ray workspace: a launch
    // waits (on the GPU) for the previous one, whatever streams the two are on
    const bool shares_workspace = rt_uses_workspace(kp);
    if (shares_workspace && !g.d_pend_ws) {
        // first frame that needs the pending-ray workspace (deep stack on a small scene)
        int wrc = ensure_workspace();
        if (wrc) return wrc;
        kp.pend_ws = g.d_pend_ws;
    }
    const bool pixel_order = wants_pixel_order(kp, p->flags);
    if (pixel_order) {
        // (the order's buffers grow with the first deep frame of a size; they are shared like the workspace)
        int orc = ensure_order_buffers((size_t)kp.cells_x * kp.my_bands * (RT_BAND_ROWS / 2));
        if (orc) return orc;
    }
    // (a wait on an event recorded on the same stream costs nothing on the GPU)
    if (shares_workspace && g.ws_busy) HIP_TRY(hipStreamWaitEvent(s, g.ws_event, 0));
    if (timed) HIP_TRY(hipEventRecord(g.ev0, s));
    if (pixel_order) {
        int orc = enqueue_pixel_order(kp, (p->flags & RT_FLAG_FAST_MATH) != 0, g.d_scene, g.d_pilot_cost, g.d_order, g.d_order_counters, s);
        if (orc) return orc;
    }
    int lrc = (p->flags & RT_FLAG_FAST_MATH) ? rt_fast::launch_render(kp, g.d_scene, d_out, s, &info)
                                             : rt_strict::launch_render(kp, g.d_scene, d_out, s, &info);
    if (lrc) return fail(RT_E_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    if (shares_workspace) {
        HIP_TRY(hipEventRecord(g.ws_event, s));
        g.ws_busy = true;
    }
    if (timed) { HIP_TRY(hipEventRecord(g.ev1, s)); g.ev_pending = true; }
    g.timing.vgprs = info.vgprs; g.timing.lds_bytes = info.lds_bytes; g.timing.pending_cap = info.pending_cap;
    return RT_OK;
}

int rt_render_device_noalloc(const rt_params *p, void *d_out, size_t out_bytes, void *stream)
{
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    if (!g.have_scene) return fail(RT_E_STATE, "no scene set (rt_set_scene)");
    int rc = validate_params(p);
    if (rc) return rc;
    size_t need = rt_output_bytes(p);
    if (!d_out && need) return fail(RT_E_ARG, "output pointer is null");
    if (out_bytes < need) return fail(RT_E_SIZE, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    KParams kp = make_kparams(p);
    if (rt_uses_workspace(kp) && rt_needed_pending(kp) > RT_NOALLOC_MAX_PENDING)
        return fail(RT_E_LIMIT, "min(stack_size, max_depth-1) = %d exceeds the %d pending-ray slots per lane of the static scratch",
                    rt_needed_pending(kp), RT_NOALLOC_MAX_PENDING);
    // queue counters and pending-ray slots come from the static array; the grid of the
    // persistent-wave kernels is bounded by the workgroups the array has slices for
    kp.work_counter = reinterpret_cast<unsigned int *>(g.scratch_ptr);
    kp.pend_ws = reinterpret_cast<float *>(g.scratch_ptr + RT_NOALLOC_COUNTER_BYTES);
    kp.pend_ws_slots = RT_NOALLOC_MAX_PENDING;
    if (kp.resident_workgroups > RT_NOALLOC_WORKGROUPS) kp.resident_workgroups = RT_NOALLOC_WORKGROUPS;
    hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : g.stream;
    // launches that use the scratch are ordered behind each other on the GPU, whatever their streams.
    // Not while `s` is being captured into a graph: waiting there on an event recorded outside the
    // capture is an error, and an event recorded inside one cannot be waited on by later plain
    // calls.  A captured call therefore leaves the library's event alone; ordering the graph's
    // replays against other users of the scratch on this GPU is the caller's job.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(s, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    const bool shares_scratch = rt_uses_workspace(kp) && !capturing;
    if (shares_scratch && g.scratch_busy) HIP_TRY(hipStreamWaitEvent(s, g.scratch_event, 0));
    LaunchInfo info{};
    int lrc = (p->flags & RT_FLAG_FAST_MATH) ? rt_fast::launch_render(kp, g.d_scene, d_out, s, &info)
                                             : rt_strict::launch_render(kp, g.d_scene, d_out, s, &info);
    if (lrc) return fail(RT_E_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    if (shares_scratch) {
        HIP_TRY(hipEventRecord(g.scratch_event, s));
        g.scratch_busy = true;
    }
    g.timing.vgprs = info.vgprs; g.timing.lds_bytes = info.lds_bytes; g.timing.pending_cap = info.pending_cap;
    return RT_OK;
}

// Output rows of the packed rows of one part, in order: bands part, part + nparts, ... of
// RT_BAND_ROWS rows each (the last band of the image may be shorter).
int rt_part_row_map(int height, int part, int nparts, int32_t *rows, int capacity)
{
    if (height <= 0 || nparts < 1 || part < 0 || part >= nparts) return fail(RT_E_ARG, "bad part/nparts");
    int bands = (height + RT_BAND_ROWS - 1) / RT_BAND_ROWS, k = 0;
    for (int b = part; b < bands; b += nparts)
        for (int r = b * RT_BAND_ROWS; r < (b + 1) * RT_BAND_ROWS && r < height; ++r) {
            if (rows && k < capacity) rows[k] = r;
            ++k;
        }
    return k;
}

// rt_init_multi mode: every part renders on its device; its bands go straight into the frame
// `d_frame` on the first device (strided device-to-device copies over xGMI, no gather buffer and no
// assemble pass); `s` (a stream of the first device) continues once all parts are in.
// Render part i of the frame on its device and copy its bands into the frame on the first device.
// The caller has made the part's device current.
static int issue_part(int i, const rt_params *p, void *d_frame, bool timed)
{
    const int np = (int)g.parts.size();
    const size_t bpp = (p->flags & RT_FLAG_OUT_F32) ? 16 : (p->flags & RT_FLAG_OUT_RGB8) ? 3 : 4;
    const size_t row_bytes = (size_t)p->width * bpp, band_bytes = row_bytes * RT_BAND_ROWS;
    const int bands = (p->height + RT_BAND_ROWS - 1) / RT_BAND_ROWS;
    const int tail_rows = p->height - (bands - 1) * RT_BAND_ROWS;          // rows of the last band (1..RT_BAND_ROWS)
    State::Part &pt = g.parts[i];
    rt_params q = *p;
    q.part = i; q.nparts = np;
    const size_t need = rt_output_bytes(&q);
    if (need > pt.d_part_cap) {
        if (pt.d_part) HIP_TRY(hipFree(pt.d_part));
        pt.d_part = nullptr; pt.d_part_cap = 0;
        HIP_TRY(hipMalloc(&pt.d_part, need ? need : 16));
        pt.d_part_cap = need ? need : 16;
    }
    const int my_bands = (bands - i + np - 1) / np;
    pt.timed = false;
    if (my_bands <= 0) { HIP_TRY(hipEventRecord(pt.done, pt.stream)); return RT_OK; }
    KParams kp = make_kparams(&q);
    kp.work_counter = pt.d_work;
    kp.resident_workgroups = pt.resident_workgroups;
    if (rt_uses_workspace(kp) && !pt.d_pend_ws) {
        size_t ws = (size_t)pt.resident_workgroups * RT_PEND_MAX * 11 * 256 * sizeof(float);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pt.d_pend_ws), ws));
    }
    kp.pend_ws = pt.d_pend_ws;
    HIP_TRY(hipStreamWaitEvent(pt.stream, g.parts[0].frame_free, 0));
    if (timed) HIP_TRY(hipEventRecord(pt.t_start, pt.stream));
    LaunchInfo info{};
    int lrc = (p->flags & RT_FLAG_FAST_MATH) ? rt_fast::launch_render(kp, pt.d_scene, pt.d_part, pt.stream, i == 0 ? &info : nullptr)
                                             : rt_strict::launch_render(kp, pt.d_scene, pt.d_part, pt.stream, i == 0 ? &info : nullptr);
    if (lrc) return fail(RT_E_HIP, "kernel launch failed on device %d: %s", pt.device, hipGetErrorString((hipError_t)lrc));
    if (timed) HIP_TRY(hipEventRecord(pt.t_kernel, pt.stream));
    // bands i, i + np, ...: contiguous in the part buffer, np bands apart in the frame
    const bool owns_tail = (bands - 1) % np == i && tail_rows != RT_BAND_ROWS;
    const int full = owns_tail ? my_bands - 1 : my_bands;
    char *dst = static_cast<char *>(d_frame) + (size_t)i * band_bytes;
    if (full > 0)
        HIP_TRY(hipMemcpy2DAsync(dst, band_bytes * np, pt.d_part, band_bytes, band_bytes, full, hipMemcpyDeviceToDevice, pt.stream));
    if (owns_tail)
        HIP_TRY(hipMemcpyAsync(dst + (size_t)full * band_bytes * np, static_cast<char *>(pt.d_part) + (size_t)full * band_bytes,
                               row_bytes * tail_rows, hipMemcpyDeviceToDevice, pt.stream));
    if (timed) { HIP_TRY(hipEventRecord(pt.t_copy, pt.stream)); pt.timed = true; }
    HIP_TRY(hipEventRecord(pt.done, pt.stream));
    if (i == 0) { g.timing.vgprs = info.vgprs; g.timing.lds_bytes = info.lds_bytes; g.timing.pending_cap = info.pending_cap; }
    return RT_OK;
}

// body of the thread that issues part i (i >= 1)
static void issuer_main(int i)
{
    Issuers &is = g_issuers;
    char err[sizeof g.err] = "";
    t_err = err;
    bool bound = hipSetDevice(g.parts[i].device) == hipSuccess;
    unsigned long long seen = 0;
    for (;;) {
        const rt_params *p; void *d_frame; bool timed;
        {
            std::unique_lock<std::mutex> lk(is.m);
            is.wake.wait(lk, [&] { return is.stop || is.generation != seen; });
            if (is.stop) return;
            seen = is.generation;
            p = is.params; d_frame = is.d_frame; timed = is.timed;
        }
        err[0] = 0;
        int rc = !p ? fail(RT_E_STATE, "no frame to issue") : bound ? issue_part(i, p, d_frame, timed) : fail(RT_E_HIP, "the issuing thread of part %d could not select device %d", i, g.parts[i].device);
        {
            std::lock_guard<std::mutex> lk(is.m);
            is.rc[i] = rc;
            is.err[i] = err;
            if (--is.pending == 0) is.finished.notify_one();
        }
    }
}

static int render_multi(const rt_params *p, void *d_frame, hipStream_t s, bool timed)
{
    const int np = (int)g.parts.size();
    if (p->nparts != 1 || p->part != 0) return fail(RT_E_ARG, "in multi-device mode the library splits the frame itself: part/nparts must be 0/1");
    Issuers &is = g_issuers;
    // the frame must not be overwritten before earlier work on `s` that reads it has run
    HIP_TRY(hipEventRecord(g.parts[0].frame_free, s));
    if (np > 1) {
        std::lock_guard<std::mutex> lk(is.m);
        is.params = p; is.d_frame = d_frame; is.timed = timed;
        is.pending = np - 1;
        is.generation++;
    }
    if (np > 1) is.wake.notify_all();
    // (no early return from here to the wait below: the issuing threads hold `p`)
    int rc = hipSetDevice(g.parts[0].device) == hipSuccess ? issue_part(0, p, d_frame, timed)
                                                            : fail(RT_E_HIP, "cannot select device %d", g.parts[0].device);
    if (np > 1) {
        std::unique_lock<std::mutex> lk(is.m);
        is.finished.wait(lk, [&] { return is.pending == 0; });
        for (int i = 1; i < np && !rc; ++i)
            if (is.rc[i]) { rc = is.rc[i]; snprintf(g.err, sizeof g.err, "%s", is.err[i].c_str()); }
    }
    HIP_TRY(hipSetDevice(g.device));
    if (rc) return rc;
    for (int i = 0; i < np; ++i) HIP_TRY(hipStreamWaitEvent(s, g.parts[i].done, 0));
    return RT_OK;
}

int rt_init_multi_devices(const int *devices, int n)
{
    if (g.inited) return fail(RT_E_STATE, "already initialised; call rt_shutdown first");
    if (!devices || n < 1 || n > 64) return fail(RT_E_ARG, "bad device list");
    int rc = rt_init(devices[0]);
    if (rc) return rc;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    g.parts.resize(n);
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= count) {
            rc = fail(RT_E_ARG, "device %d out of range (0..%d)", devices[i], count - 1);
            break;
        }
        State::Part &pt = g.parts[i];
        pt.device = devices[i];
        auto mk = [&]() -> int {
            HIP_TRY(hipSetDevice(pt.device));
            if (pt.device != g.device) {
                // the part's stream writes into the frame on the first device
                hipError_t e = hipDeviceEnablePeerAccess(g.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(RT_E_HIP, "device %d cannot access device %d: %s", pt.device, g.device, hipGetErrorString(e));
                (void)hipGetLastError();
            }
            HIP_TRY(hipStreamCreateWithFlags(&pt.stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&pt.done, hipEventDisableTiming));
            HIP_TRY(hipEventCreate(&pt.t_start));
            HIP_TRY(hipEventCreate(&pt.t_kernel));
            HIP_TRY(hipEventCreate(&pt.t_copy));
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pt.d_work), RT_SUBQUEUES * RT_SUBQUEUE_STRIDE * sizeof(unsigned int)));
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, pt.device));
            pt.resident_workgroups = prop.multiProcessorCount * 4;
            HIP_TRY(hipSetDevice(g.device));
            HIP_TRY(hipEventCreateWithFlags(&pt.frame_free, hipEventDisableTiming));
            return 0;
        };
        rc = mk();
        if (rc) break;
    }
    if (rc) { char err[sizeof g.err]; std::memcpy(err, g.err, sizeof err); rt_shutdown(); std::memcpy(g.err, err, sizeof err); return rc; }
    HIP_TRY(hipSetDevice(g.device));
    // no issuing thread is alive here: a fresh generation count, so that the new threads (which start
    // at generation 0) do not take the last frame of an earlier initialisation for a job
    g_issuers.stop = false;
    g_issuers.generation = 0;
    g_issuers.pending = 0;
    g_issuers.params = nullptr; g_issuers.d_frame = nullptr;
    g_issuers.rc.assign(n, 0);
    g_issuers.err.assign(n, std::string());
    for (int i = 1; i < n; ++i) g_issuers.threads.emplace_back(issuer_main, i);
    return RT_OK;
}

int rt_init_multi(int n_devices)
{
    if (n_devices < 1 || n_devices > 64) return fail(RT_E_ARG, "n_devices must be 1..64");
    int ids[64];
    for (int i = 0; i < n_devices; ++i) ids[i] = i;
    return rt_init_multi_devices(ids, n_devices);
}

int rt_device_count(void) { return g.inited ? (g.parts.empty() ? 1 : (int)g.parts.size()) : 0; }

int rt_render(const rt_object *objs, int n, const rt_params *p, void *out, size_t out_bytes)
{
    auto t0 = std::chrono::steady_clock::now();
    int rc = rt_set_scene(objs, n);
    if (rc) return rc;
    rc = validate_params(p);
    if (rc) return rc;
    size_t need = rt_output_bytes(p);
    if (!out && need) return fail(RT_E_ARG, "output pointer is null");
    if (out_bytes < need) return fail(RT_E_SIZE, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    rc = ensure_out(need);
    if (rc) return rc;
    rc = rt_render_device(p, g.d_out, need, g.stream, 1);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, g.d_out, need, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    g.timing.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

void *rt_host_alloc(size_t bytes)
{
    if (!g.inited) { fail(RT_E_STATE, "rt_init has not been called"); return nullptr; }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { fail(RT_E_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

int rt_host_free(void *buf)
{
    if (!buf) return RT_OK;
    HIP_TRY(hipHostFree(buf));
    return RT_OK;
}

int rt_get_timing(rt_timing *t)
{
    if (!t) return fail(RT_E_ARG, "timing pointer is null");
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    if (!g.parts.empty()) {
        // multi-device mode: the slowest part's kernel of the last timed frame
        float worst = 0.0f;
        bool any = false;
        for (int i = 0; i < (int)g.parts.size(); ++i) {
            if (!g.parts[i].timed) continue;
            float k = 0.0f, c = 0.0f;
            int rc = rt_get_part_timing(i, &k, &c);
            if (rc) return rc;
            worst = k > worst ? k : worst;
            any = true;
        }
        if (any) g.timing.kernel_ms = worst;
    }
    if (g.ev_pending) {
        HIP_TRY(hipEventSynchronize(g.ev1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, g.ev0, g.ev1));
        g.timing.kernel_ms = ms;
        g.ev_pending = false;
    }
    *t = g.timing;
    return RT_OK;
}

int rt_get_part_timing(int part, float *kernel_ms, float *copy_ms)
{
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    if (g.parts.empty()) return fail(RT_E_STATE, "not in multi-device mode (rt_init_multi)");
    if (part < 0 || part >= (int)g.parts.size() || !kernel_ms || !copy_ms) return fail(RT_E_ARG, "bad part timing arguments");
    State::Part &pt = g.parts[part];
    if (!pt.timed) return fail(RT_E_STATE, "the last frame was not timed (rt_render_device with timed != 0), or this part had no bands");
    HIP_TRY(hipSetDevice(pt.device));
    hipError_t e = hipEventSynchronize(pt.t_copy);
    if (e == hipSuccess) e = hipEventElapsedTime(kernel_ms, pt.t_start, pt.t_kernel);
    if (e == hipSuccess) e = hipEventElapsedTime(copy_ms, pt.t_kernel, pt.t_copy);
    (void)hipSetDevice(g.device);
    if (e != hipSuccess) return fail(RT_E_HIP, "part timing: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_selftest_math(uint64_t counts[6])
{
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    if (!counts) return fail(RT_E_ARG, "counts is null");
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 6 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 6 * sizeof(unsigned long long), g.stream);
    if (e == hipSuccess) e = (hipError_t)rt_strict::launch_selftest(d, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_E_HIP, "math self-test: %s", hipGetErrorString(e));
    for (int i = 0; i < 6; ++i) counts[i] = h[i];
    return RT_OK;
}

int rt_event_count(void) { return EV_COUNT; }
int rt_event_flops(int ev) { return (ev >= 0 && ev < EV_COUNT) ? k_event_flops[ev] : -1; }
const char *rt_event_name(int ev) { return (ev >= 0 && ev < EV_COUNT) ? k_event_names[ev] : ""; }

static int debug_launch(const rt_params *p, int seg, uint64_t *events, float *cap_host, size_t cap_bytes,
                        int wave_profile = 0)
{
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    if (!g.have_scene) return fail(RT_E_STATE, "no scene set (rt_set_scene)");
    int rc = validate_params(p);
    if (rc) return rc;
    KParams kp = make_kparams(p);
    kp.capture_seg = seg;
    kp.wave_profile = wave_profile;
    const size_t n_counters = wave_profile ? RT_COUNTER_SLOTS : EV_COUNT;
    size_t need_cap = 0;
    float *d_cap = nullptr;
    if (cap_host) {
        need_cap = (size_t)part_rows(p) * p->width * 4 * sizeof(float);
        if (cap_bytes < need_cap) return fail(RT_E_SIZE, "capture buffer holds %zu bytes, %zu needed", cap_bytes, need_cap);
        rc = ensure_out(need_cap);
        if (rc) return rc;
        d_cap = reinterpret_cast<float *>(g.d_out);
    }
    HIP_TRY(hipMemsetAsync(g.d_events, 0, RT_COUNTER_SLOTS * sizeof(unsigned long long), g.stream));
    int lrc = (p->flags & RT_FLAG_FAST_MATH) ? rt_fast::launch_debug(kp, g.d_scene, g.d_events, d_cap, g.stream)
                                             : rt_strict::launch_debug(kp, g.d_scene, g.d_events, d_cap, g.stream);
    if (lrc) return fail(RT_E_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    if (events)
        HIP_TRY(hipMemcpyAsync(events, g.d_events, n_counters * sizeof(unsigned long long), hipMemcpyDeviceToHost, g.stream));
    if (cap_host) HIP_TRY(hipMemcpyAsync(cap_host, d_cap, need_cap, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RT_OK;
}

int rt_count_events(const rt_params *p, uint64_t *events)
{
    if (!events) return fail(RT_E_ARG, "events pointer is null");
    return debug_launch(p, -1, events, nullptr, 0);
}

#ifdef RT_DEV_SWITCHES
// Development builds only (not part of the product ABI): the wave-level block profile of the
// instrumented kernel (tools/wave_profile.py), and the A/B switches of the launchers.
int rt_wave_profile(const rt_params *p, uint64_t *counts)       // RT_COUNTER_SLOTS entries
{
    if (!counts) return fail(RT_E_ARG, "counts pointer is null");
    return debug_launch(p, -1, counts, nullptr, 0, 1);
}
int rt_wave_profile_blocks(void) { return WP_COUNT; }
const char *rt_wave_profile_name(int i)
{
    static const char *const names[WP_COUNT] = {
#define X(n) #n,
        RT_WAVE_PROFILE_BLOCKS(X)
#undef X
    };
    return (i >= 0 && i < WP_COUNT) ? names[i] : "";
}
// per persistent wave of the last render: start, end (100 MHz ticks), loop trips, pixels finished
int rt_dev_wave_log(uint64_t *out, int max_waves)
{
    if (!g.inited || !out) return fail(RT_E_ARG, "bad wave log arguments");
    int n = g.resident_workgroups * 4 < max_waves ? g.resident_workgroups * 4 : max_waves;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, g.d_wave_log, (size_t)n * RT_WAVE_LOG_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return n;
}
int rt_dev_wave_log_clear(void)
{
    HIP_TRY(hipMemset(g.d_wave_log, 0, (size_t)g.resident_workgroups * 4 * RT_WAVE_LOG_WORDS * sizeof(unsigned long long)));
    return RT_OK;
}
void rt_dev_switch(int which, int value)
{
    rt_strict::set_dev_switch(which, value);
    rt_fast::set_dev_switch(which, value);
}
#endif

int rt_capture_segment(const rt_params *p, int seg, float *out, size_t out_bytes)
{
    if (!out) return fail(RT_E_ARG, "output pointer is null");
    if (seg < 0) return fail(RT_E_ARG, "segment index must be >= 0");
    return debug_launch(p, seg, nullptr, out, out_bytes);
}

// Kernel that scatters the gathered per-part band buffers into the final frame.
__global__ void assemble_kernel(const uint32_t *__restrict__ gathered, uint32_t *__restrict__ image, int width_dw,
                                int height, int nparts, size_t part_stride_dw)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int row = blockIdx.y;
    if (x >= width_dw || row >= height) return;
    int band = row / RT_BAND_ROWS;
    int part = band % nparts;
    int local_band = band / nparts;
    size_t src_row = (size_t)local_band * RT_BAND_ROWS + (row - band * RT_BAND_ROWS);
    image[(size_t)row * width_dw + x] = gathered[part * part_stride_dw + src_row * width_dw + x];
}

// The same for packed RGB8 parts: expands to RGBA8 (alpha 255) on the way.  One thread per
// pixel; with QUADS one thread per four pixels (3 dwords in, 4 out; needs width % 4 == 0).
__global__ void assemble_rgb8_kernel(const uint8_t *__restrict__ gathered, uint32_t *__restrict__ image, int width,
                                     int height, int nparts, size_t part_stride_bytes, int QUADS)
{
    int x = (blockIdx.x * blockDim.x + threadIdx.x) * (QUADS ? 4 : 1);
    int row = blockIdx.y;
    if (x >= width || row >= height) return;
    int band = row / RT_BAND_ROWS;
    int part = band % nparts;
    int local_band = band / nparts;
    size_t src_row = (size_t)local_band * RT_BAND_ROWS + (row - band * RT_BAND_ROWS);
    const uint8_t *src = gathered + part * part_stride_bytes + (src_row * width + x) * 3;
    uint32_t *dst = image + (size_t)row * width + x;
    if (QUADS) {
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
        uint32_t a = s4[0], b = s4[1], c = s4[2];
        uint4 v;
        v.x = (a & 0x00ffffffu) | 0xff000000u;
        v.y = (a >> 24) | ((b & 0x0000ffffu) << 8) | 0xff000000u;
        v.z = (b >> 16) | ((c & 0x000000ffu) << 16) | 0xff000000u;
        v.w = (c >> 8) | 0xff000000u;
        *reinterpret_cast<uint4 *>(dst) = v;
    } else {
        *dst = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | 0xff000000u;
    }
}

int rt_assemble_device(const void *d_gathered, void *d_image, int width, int height, int nparts,
                       int bytes_per_pixel, void *stream)
{
    if (!g.inited) return fail(RT_E_STATE, "rt_init has not been called");
    if (!d_gathered || !d_image || width <= 0 || height <= 0 || width > 32768 || height > 32768 || nparts < 1 ||
        (bytes_per_pixel != 3 && bytes_per_pixel != 4 && bytes_per_pixel != 16))
        return fail(RT_E_ARG, "bad assemble arguments");
    // every part's buffer is padded to the size of part 0 (the largest), as a gather of
    // equal-sized chunks delivers them
    rt_params p0{};
    p0.width = width; p0.height = height; p0.part = 0; p0.nparts = nparts;
    hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : g.stream;
    if (bytes_per_pixel == 3) {
        size_t stride = (size_t)part_rows(&p0) * width * 3;
        const uint8_t *src = reinterpret_cast<const uint8_t *>(d_gathered);
        uint32_t *dst = reinterpret_cast<uint32_t *>(d_image);
        // dword reads of the source and 16-byte writes of the image need aligned rows
        const bool quads = width % 4 == 0 && stride % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 4 == 0 &&
                           reinterpret_cast<uintptr_t>(dst) % 16 == 0;
        if (quads) {
            dim3 grid((width / 4 + 255) / 256, height);
            hipLaunchKernelGGL(assemble_rgb8_kernel, grid, dim3(256), 0, s, src, dst, width, height, nparts, stride, 1);
        } else {
            dim3 grid((width + 255) / 256, height);
            hipLaunchKernelGGL(assemble_rgb8_kernel, grid, dim3(256), 0, s, src, dst, width, height, nparts, stride, 0);
        }
        HIP_TRY(hipGetLastError());
        return RT_OK;
    }
    size_t stride_dw = (size_t)part_rows(&p0) * width * (bytes_per_pixel / 4);
    int width_dw = width * (bytes_per_pixel / 4);
    dim3 grid((width_dw + 255) / 256, height);
    hipLaunchKernelGGL(assemble_kernel, grid, dim3(256), 0, s, reinterpret_cast<const uint32_t *>(d_gathered),
                       reinterpret_cast<uint32_t *>(d_image), width_dw, height, nparts, stride_dw);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // extern "C"
