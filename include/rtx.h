/*
 * rtx.h -- C ABI of the MI355X render loop for malloc-man/ray-tracer.
 *
 * This is the boundary a Rust `extern "C"` block (or any FFI) binds.  The reference has no
 * FFI surface of its own; each entry point below names the Rust item whose body it replaces
 * (paths relative to the reference checkout).  Plain pointers and sizes only.
 *
 * Conventions
 *  - every call returns 0 on success; non-zero means "the reference would have panicked or
 *    the device failed" and rtx_last_error() holds the message for the calling thread;
 *  - the caller owns every input array and every output buffer; rtx_scene_upload copies
 *    what it needs, so inputs may be freed as soon as it returns;
 *  - a context may be shared by several host threads; concurrent render calls on one
 *    context run on separate HIP streams with separate scratch memory;
 *  - there is no CPU fallback: without a gfx950 device every call fails loudly.
 *
 * What "the same result" means.  Everything one World::color_at node computes (hit distance and
 * object, point, normal, over/under point, reflection vector, n1/n2, the shadow flag) has the bits
 * the reference's f64 expressions give; its lighting term agrees to an ulp (pow).  A pixel is the
 * sum of its nodes' terms, each times the product of the factors on the way up; the kernels add
 * those terms into the framebuffer with floating-point atomics in the order the hardware
 * delivers them, so a frame with reflection or refraction is reproducible from run to run, and
 * equal to the reference's recursion, only to the last few ulps (observed <= 1e-12, contract
 * 1e-9).  Frames without secondary rays are bit-reproducible.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_OK 0
#define RTX_ERR_INVALID 1   /* bad argument / unsupported scene (the reference would panic) */
#define RTX_ERR_DEVICE 2    /* HIP error, no device, out of memory */

/* src/shapes/objects.rs:14-20 (enum Shape) */
enum { RTX_SPHERE = 0, RTX_PLANE = 1, RTX_CUBE = 2, RTX_CYLINDER = 3, RTX_CONE = 4 };
/* src/surfaces/patterns.rs:13-20 (enum PatternType) */
enum { RTX_SOLID = 0, RTX_STRIPE = 1, RTX_GRADIENT = 2, RTX_RING = 3, RTX_CHECKER3D = 4, RTX_TEST = 5 };
/* Precision of the render arithmetic.  RTX_F64 is the reference's own arithmetic (see "the same result" above).
 * RTX_F32 is a fast path without a bit contract: after the reference's 8-bit conversion (src/scenes/canvas.rs:84-86)
 * at least 99.9 % of a frame's pixels are within one level of the f64 result.  One exception: objects inside
 * transformed Groups.  The reference shades such an object with a copy whose transform is rebuilt from merged
 * transformation lists (src/shapes/traits.rs:35-44) and is not the transform its ray was intersected with, so
 * over_point can lie inside the object and what the rays that start there hit is decided by the last bits of the
 * hit point, which f32 does not have; measured on randomised worlds with such groups: 95-100 % of a frame's
 * pixels, 99 % on average (tests/test_gpu_parity.py, test_f32_fast_path_on_grouped_objects).  Untransformed
 * groups and the plain objects of the same frames keep the 99.9 %. */
enum { RTX_F64 = 0, RTX_F32 = 1 };

typedef struct rtx_context rtx_context;
typedef struct rtx_scene rtx_scene;

/*
 * Flattened World (src/scenes/world.rs:6-9): one entry per Object in World.objects order,
 * which is also the tie-break order among equal t (stable sort, world.rs:98-107).
 * Matrices are 4x4 row-major doubles exactly as cached by the Object
 * (src/shapes/objects.rs:4-11, 211-215) and Pattern (src/surfaces/patterns.rs:4-10).
 */
typedef struct rtx_scene_desc {
    uint32_t n_objects;
    const uint8_t *shape_type;          /* [n] RTX_SPHERE.. */
    const double *inverse;              /* [n][16] Object.inverse_transform */
    const double *inverse_transposed;   /* [n][16] Object.inverse_transform_transposed */
    const double *ymin;                 /* [n] Cylinder/Cone min (ignored otherwise) */
    const double *ymax;                 /* [n] Cylinder/Cone max */
    const uint8_t *closed;              /* [n] Cylinder/Cone closed */
    /* Material, src/surfaces/materials.rs:4-15 */
    const double *color;                /* [n][3] */
    const double *ambient;              /* [n] */
    const double *diffuse;              /* [n] */
    const double *specular;             /* [n] */
    const double *shininess;            /* [n] */
    const double *reflective;           /* [n] */
    const double *transparency;         /* [n] */
    const double *refractive_index;     /* [n] */
    const uint8_t *casts_shadow;        /* [n] */
    /* Pattern, src/surfaces/patterns.rs:4-20 */
    const uint8_t *pattern_type;        /* [n] RTX_SOLID.. */
    const double *pattern_inverse;      /* [n][16] Pattern.inverse_transform */
    const double *pattern_color1;       /* [n][3] */
    const double *pattern_color2;       /* [n][3] */
    /* Object value-equality classes (derived PartialEq, objects.rs:3): objects that compare
     * equal alias each other in the refraction container list (world.rs:253-285).
     * NULL = every object is its own class. */
    const uint32_t *eq_class;           /* [n] or NULL */
    /* Lights, src/scenes/lights.rs:4-8.  Only lights[0] is used for shading, as in the
     * reference (world.rs:123,147); the rest are accepted and ignored. */
    uint32_t n_lights;
    const double *light_position;       /* [n_lights][3] */
    const double *light_intensity;      /* [n_lights][3] */
    /* Objects inside Groups (src/shapes/groups.rs:68-84, src/shapes/objectholders.rs:36-41), flattened
     * depth first, which is the order the nested stable sorts leave equal t in.  For such an object
     * the entries above (inverse, inverse_transposed, material, eq_class) describe the copy the
     * reference returns in its Intersections -- the child with every enclosing group's
     * transformation list merged in (combine_transforms, src/shapes/traits.rs:35-44, innermost group
     * first) -- which is what normals, patterns and the container list use.  The ray itself reaches
     * the child through chain_depth[i] + 1 matrices starting at chain_inverse[chain_first[i]]: the
     * inverse_transform of each enclosing group, outermost first, then the inverse_transform of the
     * child as it was added.  chain_depth == NULL: no groups.  At most 3 levels of nesting. */
    const uint8_t *chain_depth;         /* [n] enclosing groups (0 = plain object), or NULL */
    const uint32_t *chain_first;        /* [n] index of the object's first matrix in chain_inverse */
    const double *chain_inverse;        /* [..][16] */
} rtx_scene_desc;

/* The Camera fields ray_for_pixel reads (src/scenes/camera.rs:5-15, 147-158). */
typedef struct rtx_camera_desc {
    uint32_t hsize, vsize;
    double inverse_transform[16];
    double pixel_size, half_width, half_height;
} rtx_camera_desc;

/* Ray and kernel accounting for one call. */
typedef struct rtx_stats {
    uint64_t color_rays;      /* World::color_at invocations (primary + reflection + refraction) */
    uint64_t shadow_rays;     /* World::is_shadowed invocations (one per shaded hit) */
    uint64_t rays_per_level[8]; /* color_at invocations by recursion level (0 = primary) */
    double kernel_ms;         /* device time of the render kernels (HIP events on the call's stream) */
    double primary_kernel_ms; /* device time of the primary-ray kernel alone */
    double total_ms;          /* host wall time of the call, copies included */
    uint32_t n_launches;      /* launches that trace rays: the primary-ray kernel of every row chunk and one per recursion
                                 level of it (the small kernels between them, k_frame_begin, k_list_blocks and
                                 k_publish_counts, are not counted) */
    uint32_t n_chunks;        /* row chunks the frame was split into to bound queue memory */
    double levels_ms;         /* device time from the end of the first chunk's primary-ray kernel to the end of the
                                 frame's kernels: for a frame of one chunk, its recursion levels (one launch each) */
} rtx_stats;

/* Invoked from the calling thread after each finished row chunk with the number of pixels
 * done so far; stands in for the Arc<AtomicUsize> tracker of
 * Camera::parallel_render (src/scenes/camera.rs:173,192). */
typedef void (*rtx_progress_fn)(uint64_t pixels_done, void *user);

/* Message of the last failed call on this thread. */
const char *rtx_last_error(void);

/* One context per GPU.  device_id < 0 selects the current HIP device. */
int rtx_context_create(int device_id, rtx_context **out);
void rtx_context_destroy(rtx_context *ctx);
/* Waits for everything earlier calls on this context left in flight (rtx_render_device with a
 * stream of the caller's returns before its kernels have run) and reports what only shows once
 * they have: a secondary-ray queue that overflowed (cannot happen unless the library's own
 * sizing is wrong; the frame concerned is incomplete).  The same report is made by the next
 * render call on the context.  No counterpart in the reference. */
int rtx_context_synchronize(rtx_context *ctx);
/* Test hook: render calls made afterwards give their secondary-ray queues 1/divisor of the
 * capacity they planned, so that the overflow report above can be provoked.  1 = normal. */
int rtx_context_set_debug_queue_divisor(rtx_context *ctx, uint32_t divisor);
/* Upper bound in bytes for the secondary-ray queues of one render call (default: a sixth of
 * the device memory, at least 8 GiB); frames that could exceed it are rendered in row chunks. */
int rtx_context_set_workspace_limit(rtx_context *ctx, size_t bytes);
/* Secondary-ray queues are sized on the host for the worst case, which doubles with every level in
 * a scene that both reflects and refracts (World::reflected_color and World::refracted_color,
 * src/scenes/world.rs:160-190, both recurse): about 27 GB for a 4K frame of depth 5 whose rays
 * would fit 2 GB.  A call that returns a finished frame (no stream of the caller's, or host
 * output, or stats) and whose worst case exceeds `worst_case_bytes_above` reads the lengths of
 * each level's queues back before launching it and sizes the next level for what those rays can
 * produce at most: a host round trip per level (about 11 us) instead of the memory.  Default 8 GiB;
 * 0 = whenever the scene allows it, SIZE_MAX = never.  Scenes with bounding-sphere tests and
 * scenes whose rays cannot branch keep the plain sizing.  No counterpart in the reference. */
int rtx_context_set_measured_queues(rtx_context *ctx, size_t worst_case_bytes_above);
/* Device memory the context's render calls currently hold for secondary-ray queues. */
int rtx_context_queue_bytes(rtx_context *ctx, size_t *bytes);

/* Scenes uploaded afterwards test a ray against an object's bounding sphere before the exact
 * intersection when they hold at least this many bounded objects (spheres, cubes, cylinders with
 * finite limits); default 16, 0 = always, UINT32_MAX = never.  Results do not depend on it: the
 * tests only skip objects the reference's intersection routines would miss (no counterpart in
 * the reference, which tests every object, src/scenes/world.rs:98-107). */
int rtx_context_set_cull_threshold(rtx_context *ctx, uint32_t min_bounded_objects);
/* Scenes uploaded afterwards with bounding-sphere tests may (enabled != 0, the default) have their secondary
 * rays walk a packed copy of the sphere hierarchy kept in LDS, every lane for its own ray, when that copy fits
 * (about 11 000 objects); otherwise a wave walks the spheres in memory for its 64 rays together.  Results do
 * not depend on it (tests/test_beam_walk.py).  No counterpart in the reference. */
int rtx_context_set_lane_walk(rtx_context *ctx, int enabled);

/* Replaces the World a render borrows (src/scenes/world.rs:6-9): validates, repacks and
 * copies the scene to the device.  Fails (as the reference panics, world.rs:123,147) when
 * there are objects but no light. */
int rtx_scene_upload(rtx_context *ctx, const rtx_scene_desc *desc, rtx_scene **out);
void rtx_scene_destroy(rtx_scene *scene);
/* Which kernel specialisation the scene selected: bit 0 some object reflects, bit 1 some object is
 * transparent, bit 2 every object casts a shadow, bit 3 some object has a pattern, bit 4 the
 * bounding-sphere tests are on.  For reports; no counterpart in the reference. */
int rtx_scene_features(const rtx_scene *scene, uint32_t *bits);

/*
 * Replaces the body of Camera::parallel_render (src/scenes/camera.rs:172-201, max_depth 5),
 * Camera::preview_parallel_render (camera.rs:203-222, max_depth 2) and Camera::render
 * (camera.rs:160-170, max_depth 5): every pixel (x, y) of rows [row_begin, row_end) gets
 * World::color_at(ray_for_pixel(x, y), max_depth).
 * out_rgb is a host buffer for those rows only: (row_end-row_begin)*hsize*3 values, row-major,
 * pixel (x, y) at 3*((y-row_begin)*hsize + x) (Canvas.pixels order, src/scenes/canvas.rs:32-34);
 * double for RTX_F64, float for RTX_F32; values are not clamped.
 */
int rtx_render(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
               int precision, uint32_t row_begin, uint32_t row_end, void *out_rgb, size_t out_bytes,
               rtx_progress_fn progress, void *user, rtx_stats *stats);

/*
 * rtx_render followed by one of the reference's 8-bit conversions, done on the device so that
 * only bytes cross PCIe.  format:
 *   RTX_PIXELS_RGB8   Canvas::canvas_to_png's RGB (src/scenes/canvas.rs:80-92),
 *   RTX_PIXELS_RGBA8  Canvas::canvas_to_buffer's RGBA with alpha 255 (canvas.rs:94-104, what the
 *                     GUI displays); both take every channel as `(c * 255.999) as u8`,
 *   RTX_PIXELS_PPM8   the numbers Canvas::canvas_to_ppm prints (canvas.rs:52-71):
 *                     `(c * 255.0) as isize` clamped to 0..255, three bytes per pixel.
 * All casts truncate toward zero, saturate and send NaN to 0, as Rust's `as` does.  out holds
 * (row_end-row_begin)*hsize*(format & 0xff) bytes, row-major.
 */
#define RTX_PIXELS_RGB8 3u
#define RTX_PIXELS_RGBA8 4u
#define RTX_PIXELS_PPM8 0x103u
int rtx_render_8bit(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                    int precision, uint32_t row_begin, uint32_t row_end, uint32_t format, uint8_t *out, size_t out_bytes,
                    rtx_progress_fn progress, void *user, rtx_stats *stats);

/*
 * Same as rtx_render with the output left in device memory: d_out_rgb is a device pointer with
 * room for the same number of values, in ordinary (coarse-grained) hipMalloc memory -- the kernels
 * add into it with hardware floating-point atomics, which fine-grained or host-pinned memory
 * does not support.
 * `stream` is a hipStream_t of the caller's: the kernels are enqueued on it and the call returns
 * at once (unless stats is non-NULL: then it waits and fills stats); order anything that reads
 * d_out_rgb after the call on that stream, or call rtx_context_synchronize.  With stream == NULL
 * the call runs on a stream of the context's own and returns when the frame is complete.
 */
int rtx_render_device(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                      int precision, uint32_t row_begin, uint32_t row_end, void *d_out_rgb, size_t out_bytes,
                      void *stream, rtx_stats *stats);

/*
 * One part of a frame that several GPUs render together, left in device memory.  The image's
 * bands of band_rows rows (10 in the reference, src/scenes/camera.rs:177) are dealt out in turn:
 * band b belongs to part b mod n_parts, so that every part gets its share of the cheap rows (sky)
 * and of the dear ones (glass) -- contiguous tiles do not.  The call renders the bands of `part`
 * into d_out back to back: rtx_part_rows() rows of hsize pixels, local row L being image row
 * ((L / band_rows) * n_parts + part) * band_rows + L % band_rows.  n_parts = 1 is the whole frame.
 * format 0: three values per pixel (double or float, as rtx_render_device); an RTX_PIXELS_* value:
 * 8-bit pixels converted on the device as in rtx_render_8bit -- by the kernel that shades the
 * pixel when the frame has no secondary rays (no frame of values ever exists), by one more pass
 * over the part otherwise.  A sixth (RGBA8) or an eighth (RGB8) of the bytes of f64 values then
 * has to travel to the GPU that assembles the frame.  `stream` and `stats` as rtx_render_device.
 * Stands in for the loop over bands of Camera::parallel_render (camera.rs:181-193).
 */
uint32_t rtx_part_rows(uint32_t vsize, uint32_t band_rows, uint32_t n_parts, uint32_t part);
int rtx_render_part_device(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                           int precision, uint32_t band_rows, uint32_t n_parts, uint32_t part, uint32_t format, void *d_out,
                           size_t out_bytes, void *stream, rtx_stats *stats);

/*
 * rtx_render_device for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  The call
 * allocates and frees nothing, on the device or through any library: the counters, the per-frame
 * primary-ray tables and the secondary-ray queues that rtx_render_device keeps in memory it
 * allocates live in a file-scope device array of 256 MiB that every GPU the library runs on holds
 * from the moment its code is loaded there.  The kernels are enqueued on `stream` (NULL: the
 * null stream) and the call returns at once; there are no statistics and no progress calls.
 * Frames are split into row chunks whose queues fit the array; a frame that needs more chunks
 * than the array has counters for (32 MiB of them), or a scene whose primary-ray tables exceed
 * 8 MiB (about 60 000 objects), is refused with RTX_ERR_INVALID and nothing is enqueued: an 8K
 * frame of a scene with glass at depth 5 fits, at depth 7 it does not.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU race with
 * each other (calls on the same stream do not overlap).  Every call clears what it needs to find
 * zero, whatever the previous call left behind.
 */
int rtx_render_device_noalloc(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                              int precision, uint32_t row_begin, uint32_t row_end, void *d_out_rgb, size_t out_bytes,
                              void *stream);

/*
 * Batch form of World::color_at (src/scenes/world.rs:136-144): rays is n x 6 doubles
 * (origin xyz, direction xyz), out_rgb is n x 3 doubles (filled from the f32 result when
 * precision is RTX_F32).
 */
int rtx_trace_rays(rtx_context *ctx, const rtx_scene *scene, const double *rays, size_t n, uint32_t max_depth,
                   int precision, double *out_rgb, rtx_stats *stats);

/*
 * Per-ray view of the intermediate results of the f64 path, for parity tests against
 * intersect_world / hit_world / prepare_computations / is_shadowed / lighting
 * (src/scenes/world.rs:98-158, 242-287; src/scenes/lights.rs:43-78).
 */
typedef struct rtx_probe {
    double t;              /* hit t, or -1 when there is no hit */
    double point[3], normalv[3], over_point[3], under_point[3], reflectv[3];
    double n1, n2;
    double lighting[3];    /* lighting() term of shade_hit */
    int32_t object;        /* index of the hit object, -1 on a miss */
    int32_t inside;
    int32_t shadowed;
    int32_t pad;
} rtx_probe;
int rtx_probe_rays(rtx_context *ctx, const rtx_scene *scene, const double *rays, size_t n, rtx_probe *out);

/*
 * Self-check of the f64 kernels' division and square-root shortcuts.  The reference divides a
 * vector by its magnitude component by component (Tuple::normalize, src/matrices/tuples.rs:58-73)
 * and takes f64::sqrt in many places; the kernels share the refined reciprocal of a divisor
 * between quotients and skip the range handling of the expanded square root when the operand
 * allows, and must still produce the bits of separate IEEE divisions and roots.  Runs both ways
 * on n pseudo-random operand sets (narrow != 0: exponents near 0, where the shortcuts are taken;
 * narrow == 0: the whole f64 range with zeros, infinities, NaNs and denormals) and returns in
 * *mismatches the number of sets that differ.  Anything but 0 is a bug.
 */
int rtx_selfcheck_division(rtx_context *ctx, uint64_t seed, uint32_t n, int narrow, uint64_t *mismatches);

/*
 * Self-check of the f64 kernels' power function.  The specular term of lighting() is
 * reflect_dot_eye.powf(shininess) (src/scenes/lights.rs:70-72): the host C library's pow, whose
 * bits no device routine reproduces.  The kernels evaluate it as 2^(y log2 x) in double arithmetic
 * (a quarter of the instructions of the general-purpose device pow); the factor lies in [0, 1] and
 * colours are compared to 1e-9.  Compares the two on n pseudo-random (x, y), x in (0, 1], y in
 * [0, 2000], and returns the largest absolute difference and the largest difference relative to
 * the result.  Expected: about 2e-16 and a few 1e-13.
 */
int rtx_selfcheck_pow(rtx_context *ctx, uint64_t seed, uint32_t n, double *max_abs_diff, double *max_rel_diff);

/* Primary rays exactly as the render kernel generates them (src/scenes/camera.rs:147-158):
 * out is n x 6 doubles for pixels [first_pixel, first_pixel + n) in row-major order. */
int rtx_generate_rays(rtx_context *ctx, const rtx_camera_desc *camera, uint64_t first_pixel, size_t n, double *out);

/* (c * 255.999) as u8 per channel on the device (src/scenes/canvas.rs:84-86, 98-101):
 * d_rgb holds n_values doubles (or floats for RTX_F32) in device memory, d_out n_values bytes. */
int rtx_quantize_u8_device(rtx_context *ctx, const void *d_rgb, int precision, size_t n_values, uint8_t *d_out,
                           void *stream);

/*
 * One frame on several GPUs of a node, from one process: what the Rayon loop over bands of 10 rows
 * of Camera::parallel_render (src/scenes/camera.rs:177-193) becomes on an 8-GPU node.  The World is
 * uploaded to every GPU; band b of the image is rendered by GPU b mod n (no traffic while they
 * render); every GPU then writes its bands to their places in the frame held by the first GPU
 * (device_ids[0]) with one strided device-to-device copy over xGMI.  The frame is left there
 * (out_on_first_device != 0: `out` is memory of that GPU) or copied on to host memory.
 * format: 0 = values (double or float, 3 per pixel) or an RTX_PIXELS_* value; 8-bit pixels are
 * converted on the GPU that renders them, so a sixth (RGBA8) of the bytes travel.
 * The same device may be named more than once (a way to rehearse the path on fewer GPUs).
 * Errors of these calls are reported by rtx_multi_last_error().
 */
typedef struct rtx_multi rtx_multi;
typedef struct rtx_multi_scene rtx_multi_scene;
typedef struct rtx_multi_stats {
    uint64_t color_rays, shadow_rays;  /* filled only with RTX_MULTI_COUNT_RAYS */
    double render_ms;     /* slowest GPU: its part, device time */
    double gather_ms;     /* slowest GPU: its copy into the frame, device time */
    double total_ms;      /* host wall time of the call */
    uint64_t gather_bytes; /* bytes that left the GPUs other than the first */
    uint32_t n_devices;
    uint32_t pad;
} rtx_multi_stats;
/* count rays (makes every GPU's call wait for its kernels in turn: for reports, not for timing) */
#define RTX_MULTI_COUNT_RAYS 1u
const char *rtx_multi_last_error(void);
int rtx_multi_create(const int *device_ids, int n_devices, rtx_multi **out);
void rtx_multi_destroy(rtx_multi *multi);
int rtx_multi_device_count(const rtx_multi *multi);
int rtx_multi_scene_upload(rtx_multi *multi, const rtx_scene_desc *desc, rtx_multi_scene **out);
void rtx_multi_scene_destroy(rtx_multi_scene *scene);
int rtx_multi_render(rtx_multi *multi, const rtx_multi_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                     int precision, uint32_t format, void *out, size_t out_bytes, int out_on_first_device, uint32_t flags,
                     rtx_multi_stats *stats);

/* Device properties the benchmark reports. */
int rtx_device_info(rtx_context *ctx, char *name, size_t name_bytes, int *compute_units, int *clock_mhz);

#ifdef __cplusplus
}
#endif
#endif
