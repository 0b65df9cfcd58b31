/*
 * ptgpu.h — C ABI of the MI355X (gfx950) render hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns 0
 * on success or a negative PtStatus, and never lets an exception cross the
 * boundary.  The text of the last failure on the calling thread is available
 * from pt_last_error().
 *
 * The reference (kejcao/pathtracer @ 2024-10-08) has no FFI layer of its own.
 * The two seams this library replaces are
 *   Seam A  HitData Scene::castray(vec origin, vec direction) const   scene.cpp:18-20
 *   Seam B  void Camera::render(const Scene&, int samples, cb)        camera.cpp:171-200
 * and each declaration below cites the reference lines it stands in for.
 *
 * Threading: one host thread drives one PtScene at a time; distinct PtScene
 * objects (also on distinct devices) may be driven concurrently.
 * Streams: the calls that take a stream enqueue and return; all of them share the
 * scene's counter block and path workspace, so a call on another stream than the
 * scene's previous call first makes its stream wait (hipStreamWaitEvent) for the work
 * that call left: asynchronous calls on one PtScene run one after the other whatever
 * streams they name.
 * Current device: every entry point leaves the calling thread's current HIP device
 * as it found it.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtStatus {
  PT_OK = 0,
  PT_ERR_INVALID = -1,   /* null pointer, inconsistent sizes, malformed tree */
  PT_ERR_DEVICE = -2,    /* a HIP call failed; see pt_last_error() */
  PT_ERR_NOMEM = -3,     /* host or device allocation failed */
  PT_ERR_UNSUPPORTED = -4, /* e.g. tree deeper than the traversal stack */
  PT_ERR_NODEVICE = -5   /* no gfx950 device / code object not loadable */
} PtStatus;

/*
 * One BVH node exactly as the reference lays it out (BVH::Node, object.cpp:228-238):
 * count == 0  -> internal node, children are nodes[index_or_first] and
 *                nodes[index_or_first + 1];
 * count  > 0  -> leaf over objects[index_or_first .. index_or_first + count).
 * 32 bytes.
 * Boxes must be what the reference's builder makes them (BVH::update_box, object.cpp:296-308): the box of an
 * internal node encloses the boxes of the internal nodes and the triangles below it.  PT_TRACE_FAST relies on
 * that (it tests the lowest internal box above a triangle in place of all of them); pt_scene_upload and
 * pt_scene_check verify it, report it in PtSceneInfo.boxes_nested, and a scene whose boxes are not nested is
 * walked with PT_TRACE_EXACT whatever mode is asked for.
 */
typedef struct PtNode {
  uint32_t index_or_first;
  uint32_t count;
  float bmin[3];
  float bmax[3];
} PtNode;

/* Material fields the integrator reads (Material, object.cpp:144-156;
 * read at camera.cpp:113,122,154).  Defaults in the reference: kd=.5, ke=0. */
typedef struct PtMaterial {
  float kd[3];
  float ke[3];
} PtMaterial;

/* A decoded diffuse texture (Material::diffuse_map, object.cpp:149; Image,
 * object.cpp:31-142): RGB, 8 bits per channel, rows top to bottom as decoded,
 * no padding.  Image::at(x, y) is rgb[(y*width + x)*3 ..] / 255 (object.cpp:128-134). */
typedef struct PtTexture {
  uint32_t width, height;
  const uint8_t *rgb;
} PtTexture;

/*
 * A whole scene as the reference holds it after Scene::Scene (scene.cpp:15):
 * a top-level BVH over meshes (one Polygon per OBJ `o` group) and one private
 * BVH per mesh over its triangles (object.cpp:497-503).  Meshes and triangles
 * are given in creation ("file") order; the two `order` arrays are the
 * permutations BVH::objects holds after BVH::subdivide (object.cpp:313-361).
 * The library copies everything; the caller keeps ownership of these arrays.
 */
typedef struct PtSceneDesc {
  uint32_t n_meshes;
  uint32_t n_tris;               /* total over all meshes */
  uint32_t n_materials;
  uint32_t n_tlas_nodes;
  const PtNode *tlas_nodes;      /* [n_tlas_nodes], node 0 is the root */
  const uint32_t *tlas_order;    /* [n_meshes] slot in Scene::objects.objects -> mesh id */
  const uint32_t *blas_node_offset; /* [n_meshes+1] start of each mesh's nodes in blas_nodes */
  const PtNode *blas_nodes;      /* concatenated; index_or_first is local to the mesh */
  const uint32_t *tri_offset;    /* [n_meshes+1] first global triangle id of each mesh */
  const uint32_t *blas_order;    /* [n_tris] slot in the mesh's BVH::objects -> local triangle id */
  const float *tri_verts;        /* [n_tris][3][3]: v0.xyz v1.xyz v2.xyz (Triangle::v0/v1/v2, object.cpp:529-540) */
  const uint32_t *mesh_material; /* [n_meshes] index into materials */
  const PtMaterial *materials;   /* [n_materials] */
  /* Textured diffuse (camera.cpp:121-146).  All optional: with n_textures == 0 the three
   * pointers are ignored and every material uses its kd. */
  uint32_t n_textures;
  const PtTexture *textures;         /* [n_textures] */
  const int32_t *material_texture;   /* [n_materials] index into textures, -1 = none (diffuse_map == nullptr) */
  const float *tri_uv;               /* [n_tris][3][2]: u,v of Triangle::vt0/vt1/vt2 (object.cpp:542-553); needed
                                        when any material of the scene has a texture */
} PtSceneDesc;

typedef struct PtScene PtScene; /* opaque; owns the device copies */

typedef struct PtSceneInfo {
  uint32_t n_meshes, n_tris, n_lights;
  uint32_t n_wide_nodes;   /* 64-byte two-child device nodes of the reference-shaped tree */
  uint32_t max_depth;      /* traversal stack entries the deepest ray can need in it */
  uint32_t n_fast_nodes;   /* 128-byte four-child nodes of the fast mode's own tree */
  uint32_t fast_max_depth; /* traversal stack entries the deepest ray can need in that one */
  uint32_t n_gates;        /* internal nodes of the reference's trees (their boxes gate the fast mode's hits) */
  uint32_t device;         /* HIP device ordinal the scene lives on */
  uint64_t device_bytes;   /* bytes of HBM held by the scene */
  uint32_t boxes_nested;   /* 1: every internal box encloses what is below it (see PtNode); 0: PT_TRACE_FAST runs as PT_TRACE_EXACT */
  uint32_t n_oct_nodes;    /* 128-byte eight-child nodes of the fast tree's form for scenes outside on-chip memory (0: not built) */
} PtSceneInfo;

/* Camera state read by Camera::render (camera.cpp:49-58,161-169).  `euler` is
 * Camera::direction (radians, rotate(a,b,y) of math.cpp:130-143).  aperture and
 * focus are an extension (thin lens); aperture == 0 is the reference pinhole. */
typedef struct PtCamera {
  float origin[3];
  float euler[3];
  float fov_deg;   /* reference default 65 (camera.cpp:58) */
  int32_t vw, vh;  /* reference default 1,1 (camera.cpp:55) */
  float aperture;  /* lens radius; 0 = pinhole */
  float focus;     /* distance of the plane in focus along camera z; used when aperture > 0 */
} PtCamera;

enum { PT_SAMPLE_UNIFORM = 0, PT_SAMPLE_COSINE = 1 };
/* PT_TRACE_EXACT walks the reference's own trees in the reference's order (no pruning).
 * PT_TRACE_FAST returns the same hits from a tree of its own (four children per node, surface-area
 * heuristic), pruned by distance; a hit counts only if the ray passes the reference's box tests above
 * that triangle.  PT_TRACE_FAST_BINARY is the pruned walk over the reference-shaped trees, kept for
 * comparison (on a scene with an ill-conditioned triangle it is the exact walk).
 * The pruning modes skip boxes with a margin of 2^-17 of the scene's largest coordinate; the rounding error of
 * a box test grows with the distance of the ray's origin, so they are used for origins within 8 times that
 * coordinate of the world origin (measured: no mismatch in 600 000 rays at 10 times, the first at 100 times)
 * and every entry point sees to that by itself: pt_render and pt_render_dev run PT_TRACE_EXACT when the
 * camera stands farther away, and the ray queries (pt_trace_closest[_dev], pt_trace_any, pt_trace_hits) walk
 * every ray whose origin lies beyond that range (or is not finite) exactly, ray by ray, in a second launch.
 * One known deviation of the pruning modes: a ray lying in a triangle's plane to within rounding can make the
 * reference's triangle test report a hit at a distance unrelated to the triangle (its determinant is noise);
 * when that distance is nearer than the true closest hit, a walk that prunes by distance does not see it
 * (measured: 3 of 68 million rays aimed at vertices).  PT_TRACE_EXACT reproduces those hits too. */
enum { PT_TRACE_EXACT = 0, PT_TRACE_FAST = 1, PT_TRACE_FAST_BINARY = 2 };

/* Integrator constants.  pt_options_default() fills in the reference literals
 * (camera.cpp:52,118-119,152). */
typedef struct PtOptions {
  double rr_prob;       /* .7 as a double, as camera.cpp:118 declares it: the float draw is compared with it in
                           double (camera.cpp:119), so a draw equal to 0.7f still continues */
  int32_t rr_min_depth; /* roulette only when depth > rr_min_depth; 2 */
  int32_t max_depth;    /* 0 = unbounded (reference); k>0: at most k bounces, i.e. the vertices at depth 0..k
                           are traced and lit (k+1 closest-hit passes) and the path ends at depth k */
  int32_t sampling;     /* PT_SAMPLE_UNIFORM (reference) or PT_SAMPLE_COSINE (extension) */
  float bgcolor[3];     /* 0,0,0 */
  int32_t trace_mode;   /* PT_TRACE_FAST (default), PT_TRACE_EXACT or PT_TRACE_FAST_BINARY */
  uint32_t spp_stride;  /* sample k of this call has global index spp_begin + k*spp_stride; 1 */
  uint32_t batch_spp;   /* samples per pixel kept in flight together; 0 = library picks */
} PtOptions;

/* Work counters of one render or trace call; all are sums over rays. */
typedef struct PtCounters {
  uint64_t paths;         /* camera paths started */
  uint64_t rays_closest;  /* closest-hit queries (primary + bounce), camera.cpp:64 */
  uint64_t rays_shadow;   /* visibility queries, camera.cpp:101 */
  uint64_t node_boxes;    /* 32-byte node boxes tested by closest-hit queries (0 unless counting is on) */
  uint64_t tri_tests;     /* triangles tested by closest-hit queries (same) */
  uint64_t bounce_passes; /* depth levels processed (passes over the path queue + the levels the tail kernel adds), summed over batches */
  uint64_t shadow_node_boxes; /* node boxes tested by visibility queries (0 unless counting is on) */
  uint64_t shadow_tri_tests;  /* triangles tested by visibility queries (same) */
} PtCounters;

/* Device time of the render kernels, measured with HIP events on the stream the
 * kernels run on (see pt_scene_profile). */
typedef struct PtProfile {
  double trace_ms;        /* closest-hit kernel, summed over launches */
  double shade_ms;        /* shading kernel (light loop with its visibility rays, roulette, compaction) */
  double other_ms;        /* ray generation + resolve */
  uint64_t trace_launches;
  uint64_t shade_launches;
  double tail_ms;         /* the kernel that follows the last paths of a batch to their ends */
  uint64_t tail_launches;
  uint64_t trace_rays;    /* rays walked by the closest-hit kernel's launches (the tail kernel's are not among them) */
  double visibility_ms;   /* the visibility-walk kernel of split shading (part of shade_ms; 0 for a scene in on-chip memory, whose shading kernel walks in place) */
  uint64_t visibility_launches;
  uint64_t visibility_rays; /* visibility rays those launches walked */
} PtProfile;

const char *pt_last_error(void);

/* Number of usable devices, or a negative status. */
int pt_device_count(void);

void pt_options_default(PtOptions *opt);
void pt_camera_default(PtCamera *cam);

/* Replaces the per-frame state the reference keeps in `Scene` (scene.cpp:11-33):
 * flattens both BVH levels into one device tree, orders triangle slots so that
 * the slot index is the tie-break rank of BVH::intersect (object.cpp:265-285),
 * and builds the light table of Scene::lights() (scene.cpp:22-30) with the
 * areas of Triangle::area / Polygon::area (object.cpp:395-404,493-495).
 * Rejects a malformed description with PT_ERR_INVALID; one whose boxes are not nested (see PtNode) is
 * accepted and walked exactly. */
int pt_scene_upload(const PtSceneDesc *desc, int device, PtScene **out);
void pt_scene_free(PtScene *scene);
int pt_scene_info(const PtScene *scene, PtSceneInfo *info);
/* Host only (needs no device): runs the conversion pt_scene_upload would run, verifies the description
 * and the invariants of the trees built from it (every triangle in exactly one leaf, the fast tree's own
 * boxes nested and wide enough, references in range), checks whether the description's boxes are nested
 * (info->boxes_nested, see PtNode) and fills in the counts of `info` (device and device_bytes stay 0; may be null). */
int pt_scene_check(const PtSceneDesc *desc, PtSceneInfo *info);

/* Work counters (PtCounters.node_boxes / tri_tests): off by default.  When on,
 * separately compiled counting kernels run instead of the plain ones.  They have
 * no counterpart in the reference; HitData::hits (object.cpp:203) counts something
 * else (internal nodes above the winning leaf). */
int pt_scene_count_stats(PtScene *scene, int enable);
/* Waits for the device, returns the counters gathered since the last read and
 * clears them.  Either pointer may be null. */
int pt_scene_read_stats(PtScene *scene, uint64_t *node_boxes, uint64_t *tri_tests);
/* Per-kernel device timing of pt_render / pt_render_dev: off by default; when on, every
 * kernel launch is bracketed by HIP events.  pt_scene_read_profile returns the sums since
 * the last read and clears them. */
int pt_scene_profile(PtScene *scene, int enable);
int pt_scene_read_profile(PtScene *scene, PtProfile *out);

/* Seam A, batched: result i equals Scene::castray((ox,oy,oz)[i], (dx,dy,dz)[i])
 * (scene.cpp:18-20 -> object.cpp:265-285,406-429).  tri = global triangle id in
 * file order, -1 and t = +INF on a miss; u, v are HitData::barycentric.x/.y.
 * u and v may be null.  Host pointers; the call stages through HBM and waits. */
int pt_trace_closest(PtScene *scene, size_t n,
                     const float *ox, const float *oy, const float *oz,
                     const float *dx, const float *dy, const float *dz,
                     float *t, int32_t *tri, float *u, float *v, int mode);

/* Same with device pointers on the scene's device; enqueues on `stream`
 * (a hipStream_t, may be null) and does not wait.  In a pruning mode this is two launches: the pruning walk
 * for the rays whose origin is within its range, the exact walk for the others (see PT_TRACE_FAST); the
 * host-pointer form looks at the origins first and leaves the second launch out when none needs it. */
int pt_trace_closest_dev(PtScene *scene, size_t n,
                         const float *ox, const float *oy, const float *oz,
                         const float *dx, const float *dy, const float *dz,
                         float *t, int32_t *tri, float *u, float *v, int mode,
                         void *stream);

/* HitData::hits of Scene::castray for each ray (object.cpp:203, incremented at
 * object.cpp:281-282; the reference keeps it "for special visualizations",
 * i.e. a heat map of how deep the tree was descended): for a hit, the number
 * of internal nodes between the root and the winning triangle's leaf; for a
 * miss, the length of the chain of right children whose boxes the ray passes.
 * Host pointers. */
int pt_trace_hits(PtScene *scene, size_t n,
                  const float *ox, const float *oy, const float *oz,
                  const float *dx, const float *dy, const float *dz,
                  int32_t *hits, int mode);

/* Visibility form of Seam A as camera.cpp:101 uses it:
 * hit[i] = (Scene::castray(o,d).t != INF).  Host pointers. */
int pt_trace_any(PtScene *scene, size_t n,
                 const float *ox, const float *oy, const float *oz,
                 const float *dx, const float *dy, const float *dz,
                 uint8_t *hit, int mode);

/* Camera::point_towards for every pixel (camera.cpp:161-169): dir[(py*W+px)*3+c].
 * Host pointer. */
int pt_camera_rays(const PtCamera *cam, int W, int H, int device, float *dir_xyz);

/* Seam B (camera.cpp:171-200).  Renders samples spp_begin + k*spp_stride,
 * k in [0, spp_count), of every pixel and ADDS their radiance to
 * accum_rgb_sum[(py*W+px)*3+c] (host memory, W*H*3 floats).  The reference's
 * Camera::pixels equals sum / total_spp * 255 (camera.cpp:188-190).
 * Random numbers are a counter-based function of (seed, pixel, global sample
 * index, draw index), so any split of the sample range over calls or devices
 * gives the same per-sample radiance.  counters may be null. */
int pt_render(PtScene *scene, const PtCamera *cam, int W, int H,
              uint32_t spp_begin, uint32_t spp_count, uint64_t seed,
              const PtOptions *opt, float *accum_rgb_sum, PtCounters *counters);

/* Same with accum_rgb_sum in device memory on the scene's device; kernels are
 * enqueued on `stream`; the call returns after the last bounce pass has been
 * counted (it reads one 4-byte counter per pass) but the final add into
 * accum_rgb_sum may still be in flight on `stream`. */
int pt_render_dev(PtScene *scene, const PtCamera *cam, int W, int H,
                  uint32_t spp_begin, uint32_t spp_count, uint64_t seed,
                  const PtOptions *opt, float *accum_rgb_sum_dev,
                  PtCounters *counters, void *stream);

/* Seam B with the reference's own accumulator (camera.cpp:182-198): pixels_dev (W*H*3 floats on the scene's
 * device) holds Camera::pixels after `pass_begin` passes (all zeros for pass_begin == 0, camera.cpp:177-179)
 * and is updated by passes pass_begin .. pass_begin + pass_count - 1 with the reference's running mean,
 * pixels = (i * pixels + radiance * 255) / (i + 1), in float as vector_'s operators evaluate it
 * (math.cpp:168-183).  Pass i uses global sample index i; opt->spp_stride must be 1.  A progressive caller
 * (RenderThread::run, main.cpp:40-52) calls this once per callback interval and reads the pixels, or their
 * display conversion (pt_display_dev), back.  Otherwise like pt_render_dev. */
int pt_render_mean_dev(PtScene *scene, const PtCamera *cam, int W, int H,
                       uint32_t pass_begin, uint32_t pass_count, uint64_t seed,
                       const PtOptions *opt, float *pixels_dev, PtCounters *counters, void *stream);

/* Seam B over several devices of one node (camera.cpp:171-200; the reference's only parallelism is
 * parallel_for over OS threads, camera.cpp:14-39).  scenes[g] is a copy of the same scene on its own
 * device (pt_scene_upload once per device).  The samples spp_begin + j*spp_stride, j in [0, spp_count),
 * are dealt round-robin: device g renders j = g, g + n, ... on a host thread of its own; random numbers
 * depend on the global sample index only, so the union is the set one device would render.  The per-device
 * radiance sums are merged by ONE all-reduce (RCCL, float32 sum over 3*W*H values, loaded from librccl
 * when n > 1) and added to accum_rgb_sum (host memory).  With n == 1 the result equals pt_render's bit
 * for bit; with n > 1 it equals it up to the order of the float additions.  counters are summed over
 * the devices (bounce_passes: the maximum). */
int pt_render_multi(PtScene *const *scenes, int n_scenes, const PtCamera *cam, int W, int H,
                    uint32_t spp_begin, uint32_t spp_count, uint64_t seed, const PtOptions *opt,
                    float *accum_rgb_sum, PtCounters *counters);

/* Size of the static device buffer pt_render_dev_static works in. */
#define PT_STATIC_SCRATCH_BYTES (64u << 20)

/*
 * pt_render_dev for callers that cannot allocate: tight serving loops, harnesses that forbid
 * hipMalloc, code that must not disturb the allocator.  It has no counterpart in the reference.
 * Same arguments, same result bit for bit, but no hipMalloc / hipFree or other allocating
 * call: the counters, path queues, hit records and visibility rays live in a static device
 * array of PT_STATIC_SCRATCH_BYTES (64 MiB) that every GPU the library runs on holds until
 * the process ends.  That caps the frame: one sample of every pixel must fit, at 178 bytes
 * per pixel (+ 12 + 25 per light for scenes too large for on-chip memory), i.e. up to about
 * 375 000 pixels (e.g. 800x450; the reference's own 256x256 with 5 samples in flight);
 * pt_static_max_pixels() gives the limit for a scene, and a larger frame is rejected with
 * PT_ERR_UNSUPPORTED and a message naming the sizes.
 * Like pt_render_dev the call is not "enqueue and return": it waits for the stream once per
 * depth level (it reads the number of surviving paths to size the next launches), so it
 * cannot run inside a HIP graph capture.
 * All scenes on one GPU share that GPU's array and it keeps the last call's contents (the
 * call clears what must start at zero).  Calls on one GPU are ordered on the device by an
 * event per GPU (each waits for the previous one's kernels, whatever scene and stream that
 * one named); calls issued from several host threads at the same time still race on the host
 * side: issue them one after the other.
 */
int pt_render_dev_static(PtScene *scene, const PtCamera *cam, int W, int H,
                         uint32_t spp_begin, uint32_t spp_count, uint64_t seed,
                         const PtOptions *opt, float *accum_rgb_sum_dev,
                         PtCounters *counters, void *stream);
uint64_t pt_static_scratch_bytes(void);
/* Largest W*H pt_render_dev_static accepts for this scene in the given trace mode. */
uint64_t pt_static_max_pixels(const PtScene *scene, int trace_mode);

/* The sample dealing of pt_render_multi and of one-process-per-GPU callers, as a function: shard `shard` of
 * `n_shards` renders `*count` samples with global indices *begin + k * *stride.  The union over the shards is
 * spp_begin + j*spp_stride, j in [0, spp_count), each exactly once.  (The reference's only split of the work is
 * parallel_for over rows, camera.cpp:14-39.) */
int pt_shard_samples(uint32_t spp_begin, uint32_t spp_count, uint32_t spp_stride, uint32_t shard, uint32_t n_shards,
                     uint32_t *begin, uint32_t *count, uint32_t *stride);

/* Resolve sums to the reference's pixel values: out = sum / spp * 255
 * (camera.cpp:188-190), host memory, n floats. */
int pt_resolve(const float *accum_rgb_sum, size_t n, uint32_t spp, float *pixels);
/* The same on the device (both arrays on `device`, n floats), enqueued on `stream`. */
int pt_resolve_dev(const float *accum_rgb_sum_dev, size_t n, uint32_t spp, float *pixels_dev, int device, void *stream);
/* The display conversion of the reference's per-pass callback (main.cpp:43-47) from the sums, on the device:
 * rgb8[3*p+c] = clamp((int)(sum / spp * 255), 0, 255) with the x86-64 conversion the reference's build
 * performs (toward zero; out of int range or NaN -> INT_MIN -> 0).  n_pixels pixels, enqueued on `stream`. */
int pt_resolve_display_dev(const float *accum_rgb_sum_dev, size_t n_pixels, uint32_t spp, uint8_t *rgb8_dev,
                           int device, void *stream);
/* The same conversion from Camera::pixels themselves (as pt_render_mean_dev keeps them). */
int pt_display_dev(const float *pixels_dev, size_t n_pixels, uint8_t *rgb8_dev, int device, void *stream);

/* Device memory for hosts without a HIP toolchain of their own (the C++ Camera keeps its accumulator on the
 * device across the passes of a progressive render, camera.cpp:182-198): zeroed allocation, release, clear
 * (enqueued), and a copy to host memory that waits for `stream`.  No counterpart in the reference. */
int pt_device_alloc(int device, size_t bytes, void **out);
int pt_device_free(int device, void *ptr);
int pt_device_clear(int device, void *ptr, size_t bytes, void *stream);
int pt_device_read(int device, void *dst_host, const void *src_dev, size_t bytes, void *stream);

/* The counter-based generator, exposed so that tests can pin it:
 * Philox-4x32-10 with counter (c0,c1,c2,c3) and key (k0,k1). */
void pt_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* The roulette test of camera.cpp:118-119 for one draw, by the function the kernels use:
 * 1 when a path whose draw is `draw` goes on (the float draw is compared with the double rr_prob in double). */
int pt_roulette_pass(double rr_prob, float draw);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
