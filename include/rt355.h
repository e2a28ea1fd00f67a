/*
 * rt355.h -- C ABI of librt355.so, the MI355X (gfx950) renderer for the reference's
 * per-pixel ray-tracing path.
 *
 * The reference has no native interface for this path: its boundary is the OpenGL 3.3
 * uniform block plus one draw call made by App.render (main.py:277-300), and the result
 * is the window's RGBA8 back buffer (main.py:303).  Each entry point below names the part
 * of that boundary it replaces.  Plain C types only; every function returns 0 on success
 * or a negative RT_E_* code, with a message available from rt_last_error().  The library
 * never aborts the process (the reference calls sys.exit() on shader errors,
 * main.py:81-84, and raises on link failure, main.py:204-205).
 *
 * Threading: one host thread per process drives the library (the reference is single
 * threaded, main.py:174-178).  Calls are not re-entrant.  After rt_init the process drives
 * one GPU; after rt_init_multi it drives several from the same calls (the library then runs
 * one issuing thread of its own per additional GPU).
 */
#ifndef RT355_H
#define RT355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MAX_OBJECTS 1024   /* scene objects per frame                                 */
#define RT_MAX_LIGHT_GRID 8   /* area light: at most 8 x 8 samples                        */
#define RT_MAX_PENDING 16     /* min(stack_size, max_depth-1) must not exceed this       */
#define RT_MAX_DEPTH   64     /* max_depth must not exceed this                          */
#define RT_BAND_ROWS   16     /* rows per screen band when a frame is split over GPUs    */

enum {
    RT_OK = 0,
    RT_E_ARG = -1,        /* null pointer, bad size, bad enum                           */
    RT_E_SCENE = -2,      /* object type not 1/2, non-finite value, too many objects    */
    RT_E_LIMIT = -3,      /* stack/depth/spp beyond what the kernels are built for      */
    RT_E_STATE = -4,      /* rt_init not called / no scene set                          */
    RT_E_HIP = -5,        /* a HIP runtime call failed                                  */
    RT_E_SIZE = -6        /* output buffer too small                                    */
};

/* Mirrors `struct SceneObject` (fragment_shader.glsl:41-56) field for field, 64 bytes.
 * Filled from the scene dict list the way main.py:283-296 uploads it: a sphere leaves
 * normal/major_radius/minor_radius at 0, a torus leaves radius at 0.                     */
typedef struct rt_object {
    int32_t type;               /* 1 = sphere, 2 = torus (fragment_shader.glsl:19-20)    */
    float center[3];
    float radius;
    float normal[3];
    float major_radius;
    float minor_radius;
    float color[3];
    float reflectivity;
    float transparency;
    float refractive_index;
} rt_object;

enum {
    RT_FLAG_FAST_MATH     = 1u << 0,  /* contracted FMA + hardware rcp/rsq/sqrt/exp/log; default is
                                         the strict flavour whose arithmetic equals the reference
                                         shader's as executed by llvmpipe                           */
    RT_FLAG_NEWTON_POLISH = 1u << 1,  /* not the reference's result: polish each torus root with Newton
                                         steps on the torus' implicit function (in the torus frame, where
                                         it is well conditioned) and drop roots that do not lie on the
                                         torus; hit distances then agree with a float64 solve to ~1e-7
                                         instead of the reference's ~5e-4 (median)                    */
    RT_FLAG_BOTTOM_UP     = 1u << 2,  /* rows in GL order (row 0 = bottom); default row 0 = top     */
    RT_FLAG_OUT_F32       = 1u << 3,  /* write float RGBA (unclamped, alpha 1) instead of RGBA8     */
    RT_FLAG_NO_ACCEL      = 1u << 4,  /* always run the plain per-object loop of the reference, also
                                         on large scenes (the sphere-hierarchy traversal used there
                                         returns the same hits; this flag exists to verify that)   */
    RT_FLAG_OUT_RGB8      = 1u << 5,  /* write packed RGB8, 3 bytes per pixel (the reference's alpha
                                         is the constant 1.0, fragment_shader.glsl:674): a quarter
                                         less to move between GPUs and to the host; not with OUT_F32 */
    RT_FLAG_PIXEL_ORDER   = 1u << 6   /* deep frames of large scenes (max_depth >= 12 with 8 or more
                                         pending rays): a short pilot launch ranks the pixels by the
                                         size of their ray trees and the frame is rendered costliest
                                         pixels first.  The library does this by itself when the frame
                                         has many pixels per GPU lane; the flag forces it at any size
                                         (the frame is the same: this exists to verify that)        */
};

/* The uniforms of fragment_shader.glsl:12-15 as main.py:277-280 sets them, plus the
 * constants the shader hard-codes, as run-time parameters.                               */
typedef struct rt_params {
    float cam_to_world[16];     /* row-major 4x4, rows [x|y|view|eye] as main.py:269-274 builds
                                   it (uploaded there with transpose=GL_TRUE, main.py:279)        */
    float focal_length;         /* u_focal_length, main.py:280                                    */
    float light_pos[3];         /* u_light_pos, main.py:278                                       */
    int32_t width, height;      /* u_resolution, main.py:277                                      */
    int32_t max_depth;          /* fragment_shader.glsl:653 (4)                                   */
    int32_t stack_size;         /* fragment_shader.glsl:547 (3)                                   */
    int32_t shadow_hops;        /* fragment_shader.glsl:453 (2)                                   */
    int32_t spp_grid;           /* N: N x N stratified samples per pixel, averaged before the
                                   framebuffer clamp; 1 = the reference's single centre sample    */
    uint32_t flags;             /* RT_FLAG_*                                                      */
    int32_t part, nparts;       /* screen-space split: this call renders the bands b (RT_BAND_ROWS
                                   rows each, counted from the top of the image) with
                                   b % nparts == part, packed in band order; 0,1 = whole frame    */
    int32_t light_grid;         /* G: the light is a horizontal square of edge light_size centred
                                   on light_pos, sampled on a G x G grid (each sample with its own
                                   shadow hops; the mean of the samples' diffuse + specular terms
                                   joins the ambient term).  0 or 1 = the reference's point light
                                   (main.py:278); at most RT_MAX_LIGHT_GRID                        */
    float light_size;           /* edge of that square, >= 0                                      */
} rt_params;

typedef struct rt_timing {
    float kernel_ms;            /* last render kernel, hipEvent-timed on its stream               */
    float total_ms;             /* last rt_render call incl. upload and read-back (host clock)    */
    int32_t vgprs, lds_bytes;   /* resources of the kernel variant that ran                       */
    int32_t pending_cap;        /* pending-ray slots per lane of that variant (LDS / workspace)   */
} rt_timing;

/* Replaces context/program creation (main.py:134-161, :195-208): binds the process to
 * one GPU, creates the render stream and loads the kernels.                              */
int rt_init(int device);

/* One process, several GPUs (no counterpart: the reference draws on one GPU).  After
 * rt_init_multi(n) the same calls -- rt_set_scene, rt_render, rt_render_device with part/nparts
 * 0/1 -- use devices 0..n-1: the scene is staged on each, device i renders the RT_BAND_ROWS-row
 * bands b with b % n == i, and each device copies its bands straight into the frame on device 0
 * (strided device-to-device copies over xGMI; no gather buffer, no assemble pass), where
 * rt_render reads it back / rt_render_device leaves it in the caller's buffer, the caller's
 * stream waiting for all parts.  rt_init_multi_devices takes the device of every part explicitly;
 * a device may appear more than once (its parts then share that GPU), which is also how the
 * multi-part path is exercised on a machine with one GPU.  The event counters and
 * rt_render_device_noalloc stay single-device features of device 0.  A timed frame (timed != 0)
 * records, per part, events around its kernel and around its copy into the frame:
 * rt_get_part_timing returns the two intervals of part `part` (kernel alone; strided copy to
 * device 0 alone), and rt_get_timing's kernel_ms is then the slowest part's kernel.          */
int rt_init_multi(int n_devices);
int rt_init_multi_devices(const int *devices, int n);
int rt_device_count(void);        /* parts a frame is split into: 0 before rt_init, 1 after rt_init */
int rt_get_part_timing(int part, float *kernel_ms, float *copy_ms);

/* Host-only helper (no GPU): the output row of every packed row of part `part` of `nparts`, in
 * buffer order; returns their number (rows may be NULL to ask for it).                       */
int rt_part_row_map(int height, int part, int nparts, int32_t *rows, int capacity);

/* Replaces cleanup (main.py:350-355).                                                    */
void rt_shutdown(void);

const char *rt_last_error(void);

/* Replaces the per-object glUniform* upload (main.py:283-296): validates, derives the
 * per-object invariants (bounding radius^2, torus frame) and stages the scene on the GPU.
 * The scene stays current until the next call.                                           */
int rt_set_scene(const rt_object *objs, int n);

/* Bytes a render with these parameters writes (the packed bands of `part`).             */
size_t rt_output_bytes(const rt_params *p);

/* Replaces glUniform*(camera) + glDrawArrays (main.py:277-280, :300) for a caller that
 * owns device memory: renders the current scene into `d_out` (device pointer) on
 * `stream` (a hipStream_t, or NULL for the library's stream) without synchronising.
 * With timed != 0 the kernel is bracketed by hipEvents read back by rt_get_timing.  The library
 * keeps ONE pair of events: rt_get_timing reports the most recent timed call, so time one call at
 * a time (a second timed call issued before the first one's result has been read, on whatever
 * stream, replaces it).                                                                  */
int rt_render_device(const rt_params *p, void *d_out, size_t out_bytes, void *stream, int timed);

/* rt_render_device for callers that cannot allocate device memory around the call (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments and result as rt_render_device(p, d_out, out_bytes, stream, 0); the
 * tile-queue counters and the pending-ray slots that the persistent-wave kernels (scenes of 40+
 * objects, or min(stack_size, max_depth-1) > 3) otherwise take from buffers the library allocated
 * come from one static device array of 8 KB + RT_NOALLOC_WORKGROUPS x RT_NOALLOC_MAX_PENDING x
 * 11 x 256 x 4 bytes (46 MB) that every GPU the library is loaded on holds until the process
 * ends.  The call itself allocates and frees nothing.  Frames whose kernels need more than
 * RT_NOALLOC_MAX_PENDING pending-ray slots per lane are rejected with RT_E_LIMIT.  All calls on
 * one GPU share that array: the library orders them on the GPU in call order (one host thread, as
 * everywhere in this API); callers that get around that order (several processes cannot: the
 * array is per process) would race on it.  Under stream capture (hipStreamBeginCapture on
 * `stream`) the call records its launches into the graph and leaves the library's ordering event
 * alone: replays of that graph are ordered against other users of the array on the same GPU by
 * the caller, and every kernel variant a captured call uses must have run once before the capture
 * (its first launch on a device sets a function attribute).                                */
#define RT_NOALLOC_MAX_PENDING 4
#define RT_NOALLOC_WORKGROUPS 1024
int rt_render_device_noalloc(const rt_params *p, void *d_out, size_t out_bytes, void *stream);

/* Convenience for host callers: rt_set_scene + render + copy to `out` (host memory),
 * blocking.  This is the call that stands where App.render's draw stood (main.py:244-303). */
int rt_render(const rt_object *objs, int n, const rt_params *p, void *out, size_t out_bytes);

/* Page-locked host memory for the `out` buffer of rt_render: a frame loop that passes such a
 * buffer gets its frame back in one DMA transfer instead of a staged copy through pageable memory
 * (3840x2160 RGB8: 1.8 ms -> under 0.9 ms per frame, kernel included).  No counterpart in the
 * reference, whose frame never leaves the GPU (glfw.swap_buffers, main.py:303).  rt_host_alloc
 * returns NULL on failure (message in rt_last_error); free with rt_host_free before rt_shutdown. */
void *rt_host_alloc(size_t bytes);
int rt_host_free(void *buf);

/* After a split render: `d_gathered` holds the nparts packed buffers back to back in part
 * order (what a gather to GPU 0 delivers), each padded to the size of part 0; writes the
 * assembled frame to `d_image`.  bytes_per_pixel is that of the gathered buffers: 4 (RGBA8)
 * and 16 (float RGBA) give an image of the same format, 3 (RGB8) gives an RGBA8 image with
 * alpha 255.                                                                              */
int rt_assemble_device(const void *d_gathered, void *d_image, int width, int height,
                       int nparts, int bytes_per_pixel, void *stream);

int rt_get_timing(rt_timing *t);

/* Work accounting: runs the instrumented flavour of the same kernel over the frame and
 * returns one counter per event class (rt_event_count() of them); rt_event_flops(i) is the
 * number of scalar FP32 operations the reference shader source spends on one event i.    */
int rt_count_events(const rt_params *p, uint64_t *events);
int rt_event_count(void);
int rt_event_flops(int event);
const char *rt_event_name(int event);

/* Parity instrumentation: for the seg-th ray of every pixel-centre path in visit order
 * (0 = primary) writes 4 floats per pixel to `out` (host): object id (-1 sky, 0..n-1 scene
 * object, n ground plane, -2 no such segment), t, depth, number of segments of the path.  */
int rt_capture_segment(const rt_params *p, int seg, float *out, size_t out_bytes);

/* Host-only diagnostic (no GPU, no rt_init needed): validates a scene exactly as rt_set_scene
 * does and returns what it would stage on the GPU.  layout = { n, ntori, off_material,
 * off_torus, off_kind, total_dwords, leaf nodes, top nodes, off_top, off_leaf, off_member,
 * off_slot_id, slots per leaf, leaves per top node, 0, 0 } (offsets in dwords; node counts 0
 * when the scene is too small for the sphere hierarchy); blob (may be NULL) receives
 * total_dwords dwords followed by the 4 floats of the sphere that encloses the scene.
 * No counterpart in the reference (its scene travels as GL uniforms, main.py:283-296).   */
int rt_stage_scene_host(const rt_object *objs, int n, float *blob, size_t blob_dwords, int32_t layout[16]);

/* Self-test of the strict flavour's arithmetic (no counterpart in the reference, whose sqrt
 * and division are the CPU's): the kernels run range-restricted forms of the correctly rounded
 * sqrt, 1/sqrt and division; this compares them on the GPU with the general forms over every
 * binary32 operand of their ranges (about 1.3e10 evaluations, about a second).
 * counts[i] = operands whose results differ: sqrt, 1/sqrt, reciprocal, log2's quotient,
 * x / 3, x / 27.  All six must be 0.                                                       */
int rt_selftest_math(uint64_t counts[6]);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
