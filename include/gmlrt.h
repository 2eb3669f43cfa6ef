/* gmlrt — C ABI of the MI355X back-end for the GML `render` operator.
 *
 * The library replaces the body of the reference's renderer entry point
 *     raytracer::render(amb, lights, scene, depth, fov, w, h, filename)
 *     (/root/reference/src/raytracer.rs:206-221, called from eval_render,
 *      /root/reference/src/evaluator.rs:749-769)
 * i.e. render_pixels + trace + every Node::intersect / Primitive / Light / surface
 * call below it (raytracer.rs:44-147, primitives.rs:240-1014, lights.rs:42-120,
 * transform.rs:9-36, evaluator.rs:776-792).
 *
 * Plain C: pointers and sizes only.  All host-side numbers are f64 / i64 exactly
 * as the reference's evaluator holds them.  Every function returns 0 on success
 * or a negative gmlrt_status; nothing aborts the process.
 */
#ifndef GMLRT_H
#define GMLRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
enum gmlrt_status {
  GMLRT_OK = 0,
  GMLRT_ERR_INVALID_ARG = -1,
  GMLRT_ERR_BAD_BLOB = -2,       /* scene blob failed validation                     */
  GMLRT_ERR_NO_DEVICE = -3,      /* no usable GPU / HIP runtime failure at init      */
  GMLRT_ERR_HIP = -4,            /* a HIP call failed; see gmlrt_last_error()        */
  GMLRT_ERR_GML = -5,            /* GML front-end error (syntax, type, unbound id)   */
  GMLRT_ERR_UNSUPPORTED_SURFACE = -6, /* surface function outside the bytecode subset */
  GMLRT_ERR_IO = -7,
  GMLRT_ERR_SCENE_TOO_LARGE = -8,
  GMLRT_ERR_NOT_INITIALISED = -9,
  GMLRT_ERR_SINK = -10,          /* a gmlrt_renderer callback returned non-zero       */
  GMLRT_ERR_SURFACE = -11        /* a surface function hit a run-time error the reference aborts on
                                    (evaluator.rs:357 `divi`/`modi` by zero, :544-555 `get` out of range) for
                                    at least one shaded hit; the image and the statistics have been delivered
                                    all the same, with 0 / the array's first element in place of the failed
                                    operation                                                               */
};
const char* gmlrt_strerror(int status);
/* Text of the most recent failure on the calling thread ("" if none). */
const char* gmlrt_last_error(void);

/* ------------------------------------------------------------------ scene blob
 * The flattened scene a caller hands over: what the reference keeps as
 * Box<dyn Node> + Vec<Box<dyn Light>> + Rc<Box<dyn Fn>> surface closures
 * (primitives.rs:170-175, 321-325; lights.rs:27-100).  One contiguous buffer:
 * header, then the sections at the byte offsets the header gives.  Little endian.
 */
#define GMLRT_BLOB_MAGIC 0x54524C47u /* "GLRT" */
#define GMLRT_BLOB_VERSION 3u

enum gmlrt_prim_type { /* primitives.rs:328, 441, 576, 754, 924 */
  GMLRT_SPHERE = 0, GMLRT_CUBE = 1, GMLRT_CYLINDER = 2, GMLRT_CONE = 3, GMLRT_PLANE = 4
};
enum gmlrt_rule { /* primitives.rs:177-187 */
  GMLRT_UNION = 0, GMLRT_INTERSECT = 1, GMLRT_DIFFERENCE = 2
};
enum gmlrt_light_kind { /* lights.rs:28, 52, 77 */
  GMLRT_LIGHT_DIRECTIONAL = 0, GMLRT_LIGHT_POINT = 1, GMLRT_LIGHT_SPOT = 2
};

/* One leaf primitive with its complete object<->world pair (transform.rs:4-7).
 * Rows 0..2 of each 4x4 matrix, row-major; row 3 is always 0 0 0 1. */
typedef struct gmlrt_leaf {
  double inv_m[12];
  double m[12];
  uint32_t type;        /* gmlrt_prim_type */
  uint32_t surf_code;   /* word offset into the surface code section; GMLRT_SURF_CONST = constant surface */
  uint32_t surf_const;  /* index of this leaf's first constant in the constant pool */
  uint32_t reserved;
} gmlrt_leaf;
#define GMLRT_SURF_CONST 0xFFFFFFFFu /* constants [surf_const .. +6) = r g b kd ks n */

/* CSG tree in postfix order: one 32-bit word per node.
 *   bits 31..30 = 0: leaf,     bits 29..0 = leaf index
 *   bits 31..30 = 1: operator, bits 1..0 = gmlrt_rule,
 *                    bit 2 = left operand is a leaf, bit 3 = right operand is a leaf
 *   bits 31..30 = 2: bound (optional hint, placed BEFORE the first word of a subtree):
 *                    bits 29..15 = index into the bounds section, bits 14..0 = number of
 *                    program words of that subtree (nested bound words included).
 *                    A ray that misses the bound's sphere or its box gets an empty list
 *                    for the subtree without visiting it.  The reference has no counterpart
 *                    (it tests every primitive for every ray, primitives.rs:253-315);
 *                    a bound never changes a result as long as sphere and box each enclose
 *                    every point of the subtree's solid.  Bound words are not counted
 *                    in the tree shape: n_nodes = 2*n_leaves - 1 + n_bounds. */
#define GMLRT_NODE_LEAF(i) ((uint32_t)(i))
#define GMLRT_NODE_OP(rule, lleaf, rleaf) (0x40000000u | (uint32_t)(rule) | ((lleaf) ? 4u : 0u) | ((rleaf) ? 8u : 0u))
#define GMLRT_NODE_BOUND(idx, nwords) (0x80000000u | ((uint32_t)(idx) << 15) | (uint32_t)(nwords))
#define GMLRT_BOUND_MAX 32767u /* largest index / word count a bound word can carry */

/* World-space sphere and axis-aligned box, each enclosing the solid of one subtree
 * (with enough slack to cover the rounding of the intersection arithmetic).  The
 * sphere must be finite; a box side may be +-infinity ("no information"), and
 * lo > hi on an axis says the solid is empty.  A consumer may test either or both. */
typedef struct gmlrt_bound {
  double center[3];
  double radius;
  double lo[3];
  double hi[3];
} gmlrt_bound;

typedef struct gmlrt_light {
  uint32_t kind; /* gmlrt_light_kind */
  uint32_t reserved;
  double dir[3];   /* directional: normalize(-d) (lights.rs:36); spot: normalize(at-pos) (lights.rs:89) */
  double pos[3];   /* point / spot */
  double color[3];
  double coscutoff; /* spot: cos(cutoff degrees) (lights.rs:90) */
  double exp;       /* spot */
} gmlrt_light;

/* Surface bytecode: the (face,u,v) -> (color,kd,ks,n) function of a primitive
 * (evaluator.rs:776-792) as straight-line register code.  One 64-bit word per
 * instruction:  op[7:0] dst[15:8] a[31:16] b[47:32] c[63:48].
 * An operand with bit 15 set reads constant (surf_const + (operand & 0x7FFF));
 * otherwise it names one of 256 8-byte registers.  Registers 0,1,2 hold face
 * (i64), u, v on entry.  Integers are i64, reals f64, booleans i64 0/1.
 * Semantics of each op follow the evaluator function cited. */
enum gmlrt_sop {
  GMLRT_SOP_END = 0,
  GMLRT_SOP_MOV,   /* dst = a */
  GMLRT_SOP_ADDI,  /* evaluator.rs:315  */
  GMLRT_SOP_SUBI,  /* :480 */
  GMLRT_SOP_MULI,  /* :432 */
  GMLRT_SOP_DIVI,  /* :357  (b == 0 yields 0; the reference panics: see GMLRT_SOP_TRAP) */
  GMLRT_SOP_MODI,  /* :216, 422 */
  GMLRT_SOP_EQI,   /* :371 */
  GMLRT_SOP_LESSI, /* :400 */
  GMLRT_SOP_REAL,  /* :458 */
  GMLRT_SOP_FLOOR, /* :388 */
  GMLRT_SOP_ADDF,  /* :322 */
  GMLRT_SOP_SUBF,  /* :487 */
  GMLRT_SOP_MULF,  /* :439 */
  GMLRT_SOP_DIVF,  /* :364 */
  GMLRT_SOP_EQF,   /* :380 */
  GMLRT_SOP_LESSF, /* :411 */
  GMLRT_SOP_SQRT,  /* :474 */
  GMLRT_SOP_FRAC,  /* :394 */
  GMLRT_SOP_CLAMPF,/* :341 */
  GMLRT_SOP_SIN,   /* :464 (degrees, |r|<1e-15 snapped to 0) */
  GMLRT_SOP_COS,   /* :347 */
  GMLRT_SOP_ASIN,  /* :335 (result in degrees) */
  GMLRT_SOP_ACOS,  /* :329 */
  GMLRT_SOP_SELECT,/* dst = a ? b : c  — an `if` (:281-302) whose two arms were both evaluated */
  GMLRT_SOP_OUT,   /* result slots dst..dst+2 = a,b,c ; slots 0..2 colour, 3 kd, 4 ks, 5 n */
  GMLRT_SOP_TRAP,  /* a != 0: this evaluation fails in the reference (it panics: :357 division by zero, :544-555
                      array index out of range).  a is the condition under which the failing operation is
                      reached AND fails, so an operation on the arm of an `if` that is not taken raises nothing. */
  GMLRT_SOP__COUNT
};

typedef struct gmlrt_blob_header {
  uint32_t magic, version;
  uint64_t total_bytes;
  uint32_t n_leaves;
  uint32_t n_nodes;      /* postfix words; 2*n_leaves-1 + n_bounds */
  uint32_t n_lights;
  uint32_t n_code_words; /* 64-bit surface instruction words */
  uint32_t n_consts;     /* 8-byte constants */
  uint32_t max_stack;    /* deepest operand stack the postfix program reaches */
  uint32_t max_events;   /* per-ray event slots the traversal can need (worst case) */
  uint32_t n_bounds;     /* gmlrt_bound records (0 = no acceleration hints) */
  uint64_t off_leaves, off_nodes, off_lights, off_code, off_consts; /* byte offsets from the blob start */
  uint64_t off_bounds;
} gmlrt_blob_header;

/* ------------------------------------------------------------------ render call */
typedef struct gmlrt_stats {
  uint64_t rays_primary; /* w*h of the rows rendered (raytracer.rs:143)            */
  uint64_t rays_shadow;  /* one per (hit, light) with N.L > 0 (raytracer.rs:72-74)   */
  uint64_t rays_reflect; /* one per hit with ks > 0 && depth > 0 (raytracer.rs:100)  */
  double kernel_ms;      /* device time of the render kernel (HIP events; max over GPUs) */
  double total_ms;       /* host wall time of the call                               */
  uint32_t tiles_shared; /* 8x8-pixel tiles of this launch that four waves rendered together (0 on most launches) */
  uint32_t handout_way;  /* how the launch dealt its tiles (gmlrt_render_device's note): 0 no costs known, 1 learnt order,
                            2 image order in cost-balanced batches, 3 learnt order with shared tiles, 4 rays regrouped
                            within a block, 5 level by level (GMLRT_LEVELS=1), 6 sampled (GMLRT_SAMPLED=1); 4 to 6 only on request */
  uint64_t hits;         /* view rays (primary or reflect) that hit a surface and were shaded */
  uint32_t n_gpus;
  uint32_t surface_errors; /* non-zero: some shaded hit raised GMLRT_SOP_TRAP (the call returns GMLRT_ERR_SURFACE) */
} gmlrt_stats;

/* Operands of `render` besides the scene (evaluator.rs:749-757). */
typedef struct gmlrt_params {
  double amb[3];
  int64_t depth;
  double fov_deg;
  int64_t width, height;
  /* Row partition (image rows are independent, raytracer.rs:137-146): rows are cut
   * into tiles of tile_rows; this call renders tiles part, part+nparts, ...
   * nparts <= 1 renders every row. */
  int32_t tile_rows, part, nparts;
  int32_t reserved;
} gmlrt_params;

/* Number of rows / the row indices a partition renders, in output order. */
int64_t gmlrt_part_rows(int64_t height, int32_t tile_rows, int32_t part, int32_t nparts);
int gmlrt_part_row_index(int64_t height, int32_t tile_rows, int32_t part, int32_t nparts, int64_t* rows_out);

/* Opens `n_gpus` devices (0..n_gpus-1 of the visible set; 0 = all visible).
 * Replaces nothing in the reference (it has no device); pairs with gmlrt_shutdown. */
int gmlrt_init(int n_gpus);
void gmlrt_shutdown(void);
int gmlrt_device_count(void);

/* Drop-in for raytracer::render (raytracer.rs:206-221) with host buffers: validates
 * and uploads the blob, renders on every opened GPU (interleaved row tiles), gathers
 * into caller-owned buffers.  rgb8: w*h*3 bytes, scanline order, top row first,
 * quantised as write_ppm_file does (raytracer.rs:21-25).  rgbf64 (may be NULL): the
 * unquantised pixels the reference pushes to Renderer::push_pixel (raytracer.rs:144).
 * params->part/nparts are ignored here (the library partitions over its own GPUs). */
int gmlrt_render(const void* blob, size_t blob_bytes, const gmlrt_params* params,
                 uint8_t* rgb8, double* rgbf64, gmlrt_stats* stats);

/* Device-resident form for callers that keep data in HBM (one process per GPU).
 * A scene handle lives on the device that was current when it was created. */
typedef struct gmlrt_scene gmlrt_scene;
int gmlrt_scene_create(const void* blob, size_t blob_bytes, int device, gmlrt_scene** out);
void gmlrt_scene_destroy(gmlrt_scene* s);
/* d_rgb8 / d_rgbf64 are DEVICE pointers sized for gmlrt_part_rows(...) rows (rgbf64
 * may be NULL).  Work is queued on `hip_stream` (a hipStream_t, NULL = default
 * stream) and the call returns after the kernel has finished; stats as above.
 * A handle tunes itself to the frames it is asked for: the first call with a given
 * size, depth, field of view and partition times the pixel tiles, the next two to four
 * try other ways of handing them out (the dearest first; cost-balanced batches in image
 * order; when single tiles are dear enough to set the length of the launch, those tiles
 * rendered first by four waves each, tried twice), and from then on the fastest is used (on a frame
 * whose tiles differ much in cost the first call takes up to 1.7 times as long as the
 * later ones).  The images are the same whichever way; gmlrt_stats.handout_way names it. */
int gmlrt_render_device(gmlrt_scene* s, const gmlrt_params* params, void* d_rgb8, void* d_rgbf64,
                        void* hip_stream, gmlrt_stats* stats);

/* The final framebuffer gather of the one-process-per-GPU form: copies the packed rows a partition rendered
 * (d_rows, device memory, as gmlrt_render_device left them) to their places in a host framebuffer of `height` rows of
 * `row_bytes` each (w*3 for rgb8, w*24 for f64).  `image` is ordinary host memory; it may be a shared mapping that
 * every rank of a node has open, so that all ranks deliver their rows at once, each over its own GPU's link, and no
 * rank has to receive the others' rows first.  Calls for one device are serialised inside the library (they share
 * that device's page-locked staging).  Replaces nothing in the reference (one process, one image). */
int gmlrt_copy_rows_to_host(int device, const void* d_rows, void* image, size_t row_bytes, int64_t height,
                            int32_t tile_rows, int32_t part, int32_t nparts);

/* gmlrt_render_device for callers that cannot allocate: tight serving loops, harnesses that forbid
 * hipMalloc, code that runs between the capture and the replay of a HIP graph of its own.  It has no
 * counterpart in the reference.  Same arguments, same pixels, same ray counts; it calls no hipMalloc /
 * hipFree / hipHostMalloc or any other allocating routine.  Like gmlrt_render_device it waits for the
 * kernel (hipStreamSynchronize on `hip_stream`) and reads the ray counts back, so it cannot itself be
 * recorded into a graph: a stream that is being captured refuses the synchronisation.  Everything the kernel needs besides the scene handle (created beforehand) and the caller's
 * output buffers -- per-ray event stacks, shading contexts, counters -- lives in a file-scope __device__
 * array of GMLRT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.
 * As many rays are kept in flight as fit into that array (large CSG trees render slower than through
 * gmlrt_render_device); a scene of which not even one 256-ray block fits is refused with
 * GMLRT_ERR_SCENE_TOO_LARGE.  The hand-out order of the pixel tiles is never learnt (that needs tables).
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU race, whatever their
 * streams; the next call clears the counters the previous one left. */
#define GMLRT_NOALLOC_SCRATCH_BYTES (256u << 20)
int gmlrt_render_device_noalloc(gmlrt_scene* s, const gmlrt_params* params, void* d_rgb8, void* d_rgbf64,
                                void* hip_stream, gmlrt_stats* stats);

/* The reference's pixel sink, trait Renderer { new_image, push_pixel, done }
 * (raytracer.rs:113-117), which render() obtains from RENDERER_FACTORY
 * (raytracer.rs:183-203, 206-221; the wasm front-end installs a canvas sink through
 * set_renderer_factory, lib.rs:21-26).  gmlrt_render_to_sink is render() with that
 * contract kept: new_image(filename, w, h) once, then push_pixel once per pixel with the
 * unquantised f64 colour, in scanline order, top row first (render_pixels,
 * raytracer.rs:137-146), then done().  The reference never calls done() at this
 * snapshot (raytracer.rs:177 has no caller); here it is called, once, after the last
 * pixel; it may be NULL.  A callback that returns non-zero stops the call with
 * GMLRT_ERR_SINK.  Callbacks run on the calling thread. */
typedef struct gmlrt_renderer {
  void* user;
  int (*new_image)(void* user, const char* filename, int64_t width, int64_t height);
  int (*push_pixel)(void* user, const double rgb[3]);
  int (*done)(void* user);
} gmlrt_renderer;
int gmlrt_render_to_sink(const void* blob, size_t blob_bytes, const gmlrt_params* params, const char* filename,
                         const gmlrt_renderer* sink, gmlrt_stats* stats);

/* Replaces write_ppm_file (raytracer.rs:11-28): "P6 w h 255\n" + w*h*3 bytes. */
int gmlrt_write_ppm(const char* path, const uint8_t* rgb8, int64_t w, int64_t h);

/* ------------------------------------------------------------------ GML front-end
 * Host-side counterpart of render::render_gml (lib.rs:28-30) and of rungml's include
 * handling (rungml/src/main.rs:17-47): evaluates a GML program and, for every
 * `render` it executes, flattens the operands into a blob.  A program handle keeps
 * the blobs and operands of all render calls in execution order. */
typedef struct gmlrt_program gmlrt_program;
int gmlrt_program_open_file(const char* path, gmlrt_program** out);
int gmlrt_program_open_text(const char* gml_text, gmlrt_program** out);
void gmlrt_program_close(gmlrt_program* p);
int gmlrt_program_num_renders(const gmlrt_program* p);
/* Operands of render call `idx`; *filename points into the handle. */
int gmlrt_program_render_call(const gmlrt_program* p, int idx, gmlrt_params* params, const char** filename,
                              const void** blob, size_t* blob_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GMLRT_H */
