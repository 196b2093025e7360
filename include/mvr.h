/*
 * include/mvr.h — C ABI of libmvr_hip.so, the MI355X (gfx950) multi-view renderer.
 *
 * This is the drop-in boundary for ONE path of ajhamdi/MVTN: models/renderer.py
 * MVRenderer.forward (reference lines 173-198) -> render_meshes (65-114) /
 * render_points (116-151), plus the autograd backward the reference gets from
 * PyTorch3D (triggered at run_mvtn.py:195).  The reference is Python and binds its
 * renderer by importing PyTorch3D; a maintainer replaces those imports with the
 * ctypes stub shown in INTEGRATION.md, which calls exactly these entry points.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless marked "host";
 *   - plain pointers and sizes only: no torch / HIP types in any signature;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call
 *     only ENQUEUES work on it and returns — no allocation, no synchronisation;
 *   - flat view index n = b * n_views + v  (util.py:509-521 batch_tensor as used at
 *     models/renderer.py:79-80), N = n_meshes * n_views;
 *   - matrices are row-major; R (N,3,3) holds the camera axes as columns and
 *     X_view = X_world * R + T  (PyTorch3D row-vector convention);
 *   - images are (B, V, 3, H, W) float32, contiguous — the layout
 *     models/renderer.py:109-113 hands to its callers;
 *   - return value: MVR_OK or an error code; mvr_last_error() gives the message.
 */
#ifndef MVR_H_
#define MVR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mvr_stream_t;

enum {
  MVR_OK = 0,
  MVR_ERR_INVALID_ARG = 1,
  MVR_ERR_HIP = 2,
  MVR_ERR_WORKSPACE = 3,
  MVR_ERR_UNSUPPORTED = 4
};

/* flags for mvr_mesh_params.flags */
enum {
  MVR_CULL_BACKFACES = 1,      /* RasterizationSettings(cull_backfaces=...), models/renderer.py:98 */
  MVR_PERSPECTIVE_CORRECT = 2, /* PyTorch3D >=0.5 default for perspective cameras */
  MVR_WORKSPACE_FROM_FORWARD = 4, /* mvr_render_meshes_backward only: `workspace` is the (unmodified) buffer the
                                    matching mvr_render_meshes_forward call used -> its vertex normals are reused */
  MVR_RASTER_EXACT_ONLY = 8,   /* forward only: skip the fast rasteriser (exact coverage + provably ordered affine inverse
                                  depth) and run every strip through the exact-arithmetic kernel; same images, slower */
  MVR_BACKWARD_GATHER = 16,    /* backward only: skip the face-table kernel (per-face data fetched once per strip into LDS)
                                  and gather per covered pixel; same gradients up to summation order, slower */
  MVR_NORMALS_THREE_CROSS = 64, /* vertex normals from THREE cross products per face — (v2-v1)x(v0-v1) to vertex 1,
                                  (v0-v2)x(v1-v2) to vertex 2, (v1-v0)x(v2-v0) to vertex 0 — instead of one added to all
                                  three: the two forms PyTorch3D's Meshes._compute_vertex_normals has had; equal in exact
                                  arithmetic, ~1e-7 apart in fp32 (colours only; never coverage).  Pass it to forward AND backward */
  MVR_PROJECTION_FMA = 128,    /* world -> view as a batched matrix product accumulates it on a GPU (ascending k, fused
                                  multiply-adds: PyTorch3D's Transform3d.transform_points is a torch.bmm) instead of the closed
                                  form with every product and sum rounded (default).  The two differ in the last bit of a view
                                  coordinate: which of two COINCIDENT faces (a shell and its reversed copy — the reference ships
                                  lists of such ModelNet meshes, custom_dataset.py:92-98) wins a pixel follows; nothing else
                                  moves.  With it and fov_scale = torch's K[0][0] the pixel -> face map equals a torch-op
                                  restatement of the reference's pipeline on such a mesh at every pixel */
  MVR_KEEP_PIXEL_LISTS = 32    /* forward with pix_to_face: also leave, per (view, strip), the list of covered pixels with
                                  their winners' slots and the winners' faces in the workspace (the rasteriser has them at
                                  hand); a backward given the same workspace with MVR_WORKSPACE_FROM_FORWARD and this flag
                                  starts from them instead of re-deriving them from pix_to_face.  The workspace is larger
                                  with this flag (mvr_mesh_workspace_bytes): pass it to BOTH calls or to neither */
};

/* A batch of B triangle meshes packed back to back (what models/renderer.py:66-77 builds
 * as a PyTorch3D `Meshes`).  faces hold LOCAL vertex indices of their mesh. */
typedef struct mvr_mesh_batch {
  int32_t n_meshes;          /* B */
  int32_t total_verts;       /* sum of Vn_b */
  int32_t total_faces;       /* sum of F_b */
  int32_t max_faces;         /* max F_b (host-known; sizes the launch) */
  int32_t max_verts;         /* max Vn_b (host-known; sizes the launch) */
  int32_t reserved_;         /* keeps the pointers 8-byte aligned; set to 0 */
  const float *verts;        /* (total_verts, 3) world coordinates */
  const int32_t *faces;      /* (total_faces, 3) */
  const int32_t *vert_off;   /* (B+1) prefix sums of Vn_b */
  const int32_t *face_off;   /* (B+1) prefix sums of F_b */
  const float *vert_colors;  /* (total_verts, 3) or NULL: constant obj_color ("custom" colour, models/renderer.py:76-77,154-155) */
} mvr_mesh_batch;

/* Everything models/renderer.py:84-104 configures for the mesh path. */
typedef struct mvr_mesh_params {
  int32_t n_views;        /* V (views per mesh) */
  int32_t image_h, image_w;
  int32_t flags;          /* MVR_CULL_BACKFACES | MVR_PERSPECTIVE_CORRECT */
  int32_t n_light;        /* rows of light_dir: 1 (shared) or N (per view) */
  int32_t strip_rows;     /* rows per workgroup strip; 0 = library default */
  float fov_deg;          /* FoVPerspectiveCameras default 60 */
  float ambient, diffuse, specular, shininess; /* DirectionalLights/Materials defaults .5 .3 .2 64 */
  float obj_color[3];     /* rendering_color(), models/renderer.py:153-160 */
  float bg_color[3];      /* BlendParams(background_color), :102 */
  float z_clip;           /* PyTorch3D >= 0.5 clips faces at view depth znear / 2 = 0.5 for the perspective cameras of this
                             path (MeshRasterizer: z_clip_value) when the rasterisation is perspective-correct; 0 = no
                             clipping (PyTorch3D < 0.5).  Faces wholly nearer than the plane are culled, faces that
                             STRADDLE it are cut into one or two triangles (pytorch3d/renderer/mesh/clip.py); > 0 needs
                             MVR_PERSPECTIVE_CORRECT.  (The soft path culls, but draws straddling faces whole.) */
  float out_mean[3];      /* the trainer's Normalize (viewGCN/tools/Trainer_mvt.py:41-49) in the rasteriser's stores: the images */
  float out_std[3];       /* come out as (rgb - out_mean) / out_std per channel, the backward takes the gradient of THAT;
                             out_std[c] == 0 (a zeroed struct): channel c is stored as rendered.  Hard path only */
  /* The reference's view regulariser (ops.py:138-176: view dropout, flip, replicate-pad, random crop) as a per-view SAMPLING
   * WINDOW of the rasteriser: output pixel (row r, col c) of view n shows the source pixel (clamp(r + win_dy[n], 0, H-1),
   * flip_n(clamp(c + win_dx[n], 0, W-1))) (flip_n(x) = W-1-x where win_flip[n] != 0), times win_scale[n] (before the Normalize
   * above).  Device arrays of B*V entries; win_dx == NULL (a zeroed struct): none; win_scale may be NULL (= 1).  With
   * crop offsets (i, j) and the padding of ops.py:143: win_dy = i - pad, win_dx = j - pad.  The frames, the pixel -> face
   * map and the gradients are those of the windowed images.  Hard path only. */
  const int32_t *win_dx, *win_dy;
  const uint8_t *win_flip;
  const float *win_scale;
  int32_t win_pad;        /* subtracted from every win_dx / win_dy entry: the arrays may then hold the crop offsets (j, i) as drawn */
  float fov_scale;        /* 0: the projection scale is s = RN(1 / tan(fov_deg / 2)) evaluated in double.  > 0: s as given —
                             for a caller that wants K[0][0] of a particular PyTorch3D build (an fp32 torch.tan whose last
                             bit depends on the device: torch CPU gives 0x1.bb67b0p+0 for 60 degrees, the default here is
                             0x1.bb67aep+0) */
} mvr_mesh_params;

typedef struct mvr_points_params {
  int32_t n_clouds;       /* B */
  int32_t n_points;       /* P per cloud */
  int32_t n_views;        /* V */
  int32_t image_h, image_w;
  int32_t strip_rows;     /* 0 = default */
  int32_t points_per_pixel; /* K of PointsRasterizationSettings(points_per_pixel), :132; 0 means 1; <= 150 */
  int32_t reserved_;      /* explicit padding before the double; must be 0 */
  double radius;          /* PointsRasterizationSettings(radius), :131 — double: the compositor weight
                             1 - d2/(r*r) uses r*r evaluated in double then rounded (models/renderer.py:135-145
                             -> PointsRenderer), the rasteriser uses float(r)^2 */
  float obj_color[3];
  float bg_color[3];
  float out_mean[3];      /* as in mvr_mesh_params: the images come out as (rgb - out_mean) / out_std per channel */
  float out_std[3];       /* (the trainer's Normalize in the rasteriser's stores); out_std[c] == 0: as rendered */
  /* per-view sampling window, exactly as in `mvr_mesh_params`: the view regulariser of ops.py:138-176 rendered by the point
   * rasterisers, K = 1 and K > 1, forward and backward; device arrays of B*V entries; win_dx == NULL: none */
  const int32_t *win_dx, *win_dy;
  const uint8_t *win_flip;
  const float *win_scale;
  int32_t win_pad;        /* as in mvr_mesh_params */
  int32_t reserved2_;     /* explicit padding; must be 0 */
} mvr_points_params;

/* ABI generation of this header: bumped whenever a struct layout or an entry point's contract changes (101: fov_scale in
 * mvr_mesh_params, flag 64, 8-entry arrays of mvr_profile_end; 102: mvr_mesh_fast_stats; 103: flag 128 = MVR_PROJECTION_FMA, the mesh workspace carries the exact settle's candidates inside the
 * strips' list regions — mvr_mesh_workspace_bytes answers 30 % less, and a forward's workspace is not readable by a 102 backward).  mvr_version() returns the value the library was built with;
 * a loader must refuse a library whose value differs (mvtn_amd/_lib.py does). */
#define MVR_ABI_VERSION 103
int mvr_version(void);
const char *mvr_last_error(void);
/* Hash of the sources (and compiler flags) this library was built from, as mvtn_amd/build.py computes it: lets a loader
 * recognise a library that does not belong to the tree beside it without a side file. */
const char *mvr_source_hash(void);

/* Measurement aid (bench.py): between begin and end, the dominant kernel of each path is bracketed by
 * HIP events recorded on the stream it is launched on.  mvr_profile_end synchronises on those events and
 * returns, per kind {0: mesh raster+shade forward, 1: points raster+shade forward, 2: mesh backward, 3: soft mesh
 * forward, 4: soft mesh backward, 5: mesh pre-pass, 6-7: unused}, the summed kernel time in ms and the number of launches
 * (both host arrays of length 8).  The environment variable MVR_PROF_MASK (read by mvr_profile_begin; bit k = kind k) selects
 * the kinds; default 0xDF: all but the pre-pass, whose events would sit between the launches the mesh forward's timing brackets. */
/* Self-test of the rasteriser's correctly rounded reciprocal (hardware v_rcp_f32 + two Newton steps) against the
 * IEEE division 1.0f / d for every float whose bit pattern lies in [lo_bits, hi_bits]; *n_bad (device) receives the
 * number of mismatches, *first_bad (device) one offending bit pattern.  Used by the GPU parity tests only. */
int mvr_selftest_rcp(uint32_t lo_bits, uint32_t hi_bits, uint32_t *n_bad, uint32_t *first_bad, mvr_stream_t stream);

int mvr_profile_begin(void);
int mvr_profile_end(double *total_ms, int32_t *launches);

/* ---- cameras ----------------------------------------------------------- */

/* look_at_view_transform(dist, elev, azim) in degrees, at=origin, up=+Y
 * (models/renderer.py:79-80,122-123) fused with the validity test of
 * check_valid_rotation_matrix (util.py:403-420; ops.py:156-165 is the caller):
 * view_valid[n] = 1 iff allclose(R R^T, I, atol=1e-6) and allclose(det R, 1);
 * status[0] = 0, status[1] = number of invalid views.  view_valid / status may be NULL. */
int mvr_look_at(const float *azim, const float *elev, const float *dist, int32_t n,
                float *R, float *T, int32_t *view_valid, int32_t *status /* int32[2] */,
                mvr_stream_t stream);

/* The same, fused with the remedy loop of check_and_correct_rotation_matrix (ops.py:156-165,
 * called at models/renderer.py:81-82,124-125 with nb_trials = EXAHSTION_LIMIT = 20): while ANY view
 * is invalid, ALL views are re-drawn as look_at(azim + 180 U, elev + 90 U', dist), U, U' ~ U[0,1)
 * from a counter-based generator keyed by `seed`; the loop stops when every view is valid or a
 * re-draw numbered > max_trials is still invalid (where the reference calls sys.exit).
 * Runs entirely on the device: no host synchronisation (the reference syncs on every forward).
 *   azim_used / elev_used (n) out: the angles R, T were finally built from (needed by the backward);
 *   status[0] = re-draws performed, status[1] = views still invalid (non-zero <=> "Remedy did not work"). */
int mvr_look_at_checked(const float *azim, const float *elev, const float *dist, int32_t n,
                        int32_t max_trials, uint64_t seed, float *R, float *T, float *azim_used,
                        float *elev_used, int32_t *view_valid, int32_t *status /* int32[2] */,
                        mvr_stream_t stream);

/* autograd backward of mvr_look_at: (dL/dR (n,9), dL/dT (n,3)) -> dL/d(azim, elev, dist) in
 * the units of the inputs (degrees).  grad_R or grad_T may be NULL (treated as zero). */
int mvr_look_at_backward(const float *azim, const float *elev, const float *dist, int32_t n,
                         const float *grad_R, const float *grad_T, float *grad_azim,
                         float *grad_elev, float *grad_dist, mvr_stream_t stream);

/* camera_position_from_spherical_angles (models/renderer.py:168-169, "relative" light). */
int mvr_camera_position(const float *azim, const float *elev, const float *dist, int32_t n,
                        float *position, mvr_stream_t stream);
int mvr_camera_position_backward(const float *azim, const float *elev, const float *dist,
                                 int32_t n, const float *grad_position, float *grad_azim,
                                 float *grad_elev, float *grad_dist, mvr_stream_t stream);

/* ---- mesh hand-over: models/renderer.py:66-77 ---------------------------- */

/* Where the arrays of a `mvr_mesh_batch` sit inside ONE block of `bytes` bytes (every section 256-byte
 * aligned), so that B host meshes reach HBM in a single copy (the reference issues 2*B `.cuda()` copies,
 * models/renderer.py:67-68, and packs on the device, :70-77). */
typedef struct mvr_pack_layout {
  int32_t n_meshes, total_verts, total_faces, max_verts, max_faces, reserved_;
  uint64_t verts_at, faces_at, vert_off_at, face_off_at; /* byte offsets inside the block */
  uint64_t bytes;                                        /* size of the block */
} mvr_pack_layout;

/* HOST functions (no GPU call).  n_verts / n_faces: (B) sizes of the meshes. */
int mvr_pack_meshes_layout(int32_t n_meshes, const int32_t *n_verts, const int32_t *n_faces,
                           mvr_pack_layout *layout);
/* verts[b]: (n_verts[b],3) f32, faces[b]: (n_faces[b],3) i64 (the dtype `faces_list()` hands over,
 * models/renderer.py:68), both host memory.  Writes verts, faces narrowed to i32 (LOCAL indices, checked
 * against [0, n_verts[b])), and both prefix-sum arrays into `block` (host memory, ideally pinned);
 * the caller copies the block to HBM and points a mvr_mesh_batch into it. */
int mvr_pack_meshes_host(const mvr_pack_layout *layout, const float *const *verts, const int32_t *n_verts,
                         const int64_t *const *faces, const int32_t *n_faces, void *block);

/* ---- soft mesh path (models/renderer.py:3-12 imports SoftPhongShader / SoftSilhouetteShader; :42,:89-93 hold
 * blur_radius / faces_per_pixel; the reference leaves them at 0 / 1 with the hard shader) --------------------------- */
enum { MVR_SHADER_SOFT_PHONG = 0, MVR_SHADER_SOFT_SILHOUETTE = 1 };
typedef struct mvr_soft_params {
  float blur_radius;                 /* RasterizationSettings.blur_radius (squared NDC distance), >= 0 */
  float sigma, gamma;                /* BlendParams (PyTorch3D defaults 1e-4, 1e-4) */
  int32_t faces_per_pixel;           /* K in [1, 64] */
  int32_t shader;                    /* MVR_SHADER_SOFT_PHONG: phong_shading of every fragment + softmax_rgb_blend;
                                        MVR_SHADER_SOFT_SILHOUETTE: sigmoid_alpha_blend */
  int32_t clip_barycentric_coords;   /* PyTorch3D: None -> True when blur_radius > 0 */
} mvr_soft_params;

size_t mvr_mesh_soft_workspace_bytes(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const mvr_soft_params *soft);

/* images: (N, 4, H, W) RGBA planar (N = B V, n = b V + v; alpha = 1 - prod_k (1 - sigmoid(-dist_k / sigma)));
 * pix_to_face: (N, H, W, K) LOCAL face indices of the K nearest candidate faces per pixel in ascending (z, face) order,
 * -1 = none (required: the backward differentiates exactly these fragments).  Any 4-byte alignment is accepted; with
 * faces_per_pixel = 4 a 16-byte aligned array (what every device allocator returns) lets the kernel read a pixel's list in one load. */
int mvr_render_meshes_soft_forward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const mvr_soft_params *soft,
                                   const float *R, const float *T, const float *light_dir, float *images, int32_t *pix_to_face,
                                   void *workspace, size_t workspace_bytes, mvr_stream_t stream);
/* grad_images: (N, 4, H, W).  The fragment selection is a constant; everything else (colours, depths in the softmax,
 * signed distances in the coverage probabilities) is differentiated.  grad_light: (N, 3) or NULL. */
int mvr_render_meshes_soft_backward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const mvr_soft_params *soft,
                                    const float *R, const float *T, const float *light_dir, const int32_t *pix_to_face,
                                    const float *grad_images, float *grad_R, float *grad_T, float *grad_light, void *workspace,
                                    size_t workspace_bytes, mvr_stream_t stream);

/* ---- mesh path: models/renderer.py:65-114 ------------------------------ */

/* Scratch the mesh calls need (vertex attributes, projected vertices, backward partials). */
size_t mvr_mesh_workspace_bytes(const mvr_mesh_batch *meshes, const mvr_mesh_params *params);

/* How many faces of the LAST forward that used `workspace` straddled the z-clip plane and were cut there (summed over the
 * views; 0 for unit-normalised objects at distance >= 1.5).  Views with such faces take the exact rasteriser.  In the soft
 * path they are rasterised whole (as PyTorch3D < 0.5 does): there the count tells whether a forward was clip-exact.
 * count: 1 device int. */
int mvr_mesh_clip_straddlers(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const void *workspace, size_t workspace_bytes,
                             int32_t *count, mvr_stream_t stream);

/* Diagnostics: how many (view, strip) tiles of the LAST forward that used `workspace` the fast rasteriser could not
 * decide and handed to the exact-arithmetic kernel (coincident / intersecting faces, extreme depths).  Only meaningful
 * after a forward that ran the fast rasteriser (perspective-correct, no MVR_RASTER_EXACT_ONLY).  count: 1 device int. */
int mvr_mesh_redo_count(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const void *workspace, size_t workspace_bytes,
                        int32_t *count, mvr_stream_t stream);

/* More of the same, counts: 4 device ints: [0] the redo count above; [1] the (pixel, face) CANDIDATES the fast rasteriser recorded
 * where it could not prove the depth order and handed to the reference's exact test inside its own workgroup (up to 4 per pixel
 * pair and meeting; per strip capped at 32 767) — not a pixel count; [2] strip-list entries (a face is listed once per strip it
 * reaches) that a WHOLE WAVE rasterised: the "fat" ones among the entries wider than 16 pixels (a wide entry none of whose rows
 * is longer than 8 pixel pairs runs one lane per row and is not counted); per strip capped at 16 383; [3] strip-list entries in
 * all.  A strip the exact body redid contributes to [1] / [2] what the fast path had counted before it gave up (0 when it never
 * started). */
int mvr_mesh_fast_stats(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const void *workspace, size_t workspace_bytes,
                        int32_t *counts, mvr_stream_t stream);

/* World-space area-weighted vertex normals (Meshes.verts_normals_packed, used by
 * HardPhongShader at :102).  normals: (total_verts, 3).  Uses `workspace` as scratch. */
int mvr_vertex_normals(const mvr_mesh_batch *meshes, float *normals, void *workspace,
                       size_t workspace_bytes, mvr_stream_t stream);

/* MeshRenderer(MeshRasterizer(blur_radius=0, faces_per_pixel=1), HardPhongShader) over the
 * V-times extended batch (:89-107) + the re-layout of :109-113.
 *   R (N,9), T (N,3): cameras;  light_dir (n_light,3): DirectionalLights direction (:190-191)
 *   images (B,V,3,H,W) out;  pix_to_face (N,H,W) out or NULL: LOCAL face index, -1 = background
 *   (must be non-NULL if mvr_render_meshes_backward will be called). */
int mvr_render_meshes_forward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params,
                              const float *R, const float *T, const float *light_dir,
                              float *images, int32_t *pix_to_face, void *workspace,
                              size_t workspace_bytes, mvr_stream_t stream);

/* autograd backward of the above w.r.t. the cameras and the light (vertices carry no
 * gradient in MVTN: they come from the DataLoader).  pix_to_face is treated as a constant,
 * exactly as PyTorch3D's hard rasteriser/blend does (no coverage gradient).
 *   grad_images (B,V,3,H,W) in;  grad_R (N,9), grad_T (N,3) out;
 *   grad_light (N,3) out or NULL (only meaningful when n_light == N). */
int mvr_render_meshes_backward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params,
                               const float *R, const float *T, const float *light_dir,
                               const int32_t *pix_to_face, const float *grad_images,
                               float *grad_R, float *grad_T, float *grad_light,
                               void *workspace, size_t workspace_bytes, mvr_stream_t stream);

/* ---- the whole mesh path of MVRenderer.forward in one call --------------------------------
 * models/renderer.py:79-113: look_at_view_transform (+ check_and_correct_rotation_matrix on the device when
 * max_trials >= 0, see mvr_look_at_checked) -> lights -> MeshRenderer(MeshRasterizer, HardPhongShader).  Same kernels as
 * mvr_look_at_checked + mvr_camera_position + mvr_render_meshes_forward, fewer launches and one host call.
 *   light_dir (params->n_light,3), or NULL for the reference's "relative" light_direction (models/renderer.py:168-169):
 *   the camera position of every REQUESTED (azim, elev, dist) is written to light_used (N,3) and used as the light;
 *   params->n_light must then be N.  azim_used / elev_used (N) receive the angles R, T were finally built from
 *   (= the inputs unless the remedy re-drew them); view_valid / status as in mvr_look_at_checked (may be NULL). */
int mvr_render_mesh_views_forward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params,
                                  const float *azim, const float *elev, const float *dist,
                                  int32_t max_trials, uint64_t seed, const float *light_dir,
                                  float *R, float *T, float *azim_used, float *elev_used, float *light_used,
                                  int32_t *view_valid, int32_t *status, float *images, int32_t *pix_to_face,
                                  void *workspace, size_t workspace_bytes, mvr_stream_t stream);

/* Its backward, (d azim, d elev, d dist) straight from grad_images: mvr_render_meshes_backward followed by
 * mvr_look_at_backward at (azim_used, elev_used) and, if relative_light, mvr_camera_position_backward at
 * (azim, elev), summed.  `light` is what the forward used (light_dir or light_used).  Pass the forward's workspace
 * with MVR_WORKSPACE_FROM_FORWARD to reuse its vertex attributes. */
int mvr_render_mesh_views_backward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params,
                                   const float *azim, const float *elev, const float *dist,
                                   const float *azim_used, const float *elev_used, const float *R, const float *T,
                                   const float *light, int32_t relative_light, const int32_t *pix_to_face,
                                   const float *grad_images, float *grad_azim, float *grad_elev, float *grad_dist,
                                   void *workspace, size_t workspace_bytes, mvr_stream_t stream);

/* ---- point-cloud path: models/renderer.py:116-151 ---------------------- */

size_t mvr_points_workspace_bytes(const mvr_points_params *params);

/* Pointclouds.extend(V).scale_(1/dist) -> OpenGLOrthographicCameras(R,T) ->
 * PointsRasterizer(radius, points_per_pixel=K) -> NormWeightedCompositor(background).
 *   points (B,P,3); point_colors (B,P,3) or NULL (constant obj_color); dist (N);
 *   images (B,V,3,H,W) out; pix_to_point (N,H,W,K) out or NULL: LOCAL point indices of the K nearest
 *   points of the pixel in ascending (z, index), -1 = empty slot (empties are at the end). */
int mvr_render_points_forward(const mvr_points_params *params, const float *points,
                              const float *point_colors, const float *R, const float *T,
                              const float *dist, float *images, int32_t *pix_to_point,
                              void *workspace, size_t workspace_bytes, mvr_stream_t stream);

/* The forward of the point path as ONE call (the look-at of models/renderer.py:122-125 runs inside the projection
 * pre-pass: two launches instead of three).  azim / elev / dist: (N) degrees; max_trials / seed, azim_used /
 * elev_used / view_valid / status: as mvr_look_at_checked; R (N,9), T (N,3) are OUTPUTS.  At most 2048 views per call
 * (MVR_ERR_UNSUPPORTED beyond: use mvr_look_at_checked + mvr_render_points_forward).  Everything else as
 * mvr_render_points_forward. */
int mvr_render_point_views_forward(const mvr_points_params *params, const float *points, const float *point_colors,
                                   const float *azim, const float *elev, const float *dist, int32_t max_trials,
                                   uint64_t seed, float *R, float *T, float *azim_used, float *elev_used,
                                   int32_t *view_valid, int32_t *status, float *images, int32_t *pix_to_point,
                                   void *workspace, size_t workspace_bytes, mvr_stream_t stream);

/* backward w.r.t. cameras and dist.  The compositor's weight gradient is PyTorch3D's
 * (f_k S - sum_j w_j f_j) / S^2 with S = max(sum_j w_j, 1e-4): with points_per_pixel = 1 it is exactly
 * zero except where the clamp is active.  pix_to_point is the forward's (N,H,W,K). */
int mvr_render_points_backward(const mvr_points_params *params, const float *points,
                               const float *point_colors, const float *R, const float *T,
                               const float *dist, const int32_t *pix_to_point,
                               const float *grad_images, float *grad_R, float *grad_T,
                               float *grad_dist, void *workspace, size_t workspace_bytes,
                               mvr_stream_t stream);

/* ---- the step after the renderer: ops.py:168-176 (run_mvtn.py:186-187) ---- */

/* regualarize_rendered_views = dropout2d over whole views (ops.py:170-171: on the 5-D (B,V,C,H,W) tensor the keep
 * mask is (B,V,1,1,1)) + applied_transforms (ops.py:138-146) per image: RandomHorizontalFlip (util.py:168-196),
 * ReplicationPad2d(pad), RandomCrop(H) (util.py:199-242) — applied in ONE pass:
 *   out[n,c,i,j] = scale[n] * images[n,c, clamp(i + crop_i[n] - pad), flip_n(clamp(j + crop_j[n] - pad))]
 * images, out: (n_images, channels, h, w) f32, distinct buffers.  Per image n (memory order of `images`):
 *   scale  (n_images) f32: 0 for a dropped view, 1/(1-p) otherwise; NULL = no dropout
 *   flip   (n_images) u8 : non-zero = mirror the columns;           NULL = no flips
 *   crop_i, crop_j (n_images) i32 in [0, 2*pad]: offsets of the crop window in the padded image (both or neither;
 *          NULL = centred window = no shift).
 * The caller draws mask / flips / offsets (torch generators, in the reference's order); the kernels apply them.
 * Backward: grad_images[n,c,y,x] = scale[n] * sum of grad_out over the output pixels that read (y,x), summed in a
 * fixed order (bit-reproducible; the reference's replication-pad backward uses atomics). */
int mvr_regularize_views_forward(const float *images, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                 const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                 const int32_t *crop_j, int32_t pad, float *out, mvr_stream_t stream);
int mvr_regularize_views_backward(const float *grad_out, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                  const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                  const int32_t *crop_j, int32_t pad, float *grad_images, mvr_stream_t stream);

/* The same pass with the per-channel `Normalize` of the reference's trainer fused behind it
 * (viewGCN/tools/Trainer_mvt.py:41-49: (x - 0.456) / 0.225 on every channel, applied right after `mvctosvc`, ops.py:283,
 * which is a pure reshape of the contiguous (B, V, C, H, W) block to (B V, C, H, W) and costs nothing here):
 *   out = (regularised - mean[c]) / std[c]      (true division: the bits of the torch expression)
 * mean, std: `channels` (<= 4) HOST floats each, or both NULL.  scale / flip / crop_i / crop_j may all be NULL: then
 * this is the stand-alone Normalize (one read + one write of every frame). */
int mvr_regularize_normalize_views_forward(const float *images, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                           const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                           const int32_t *crop_j, int32_t pad, const float *mean, const float *std_,
                                           float *out, mvr_stream_t stream);
int mvr_regularize_normalize_views_backward(const float *grad_out, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                            const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                            const int32_t *crop_j, int32_t pad, const float *mean, const float *std_,
                                            float *grad_images, mvr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVR_H_ */
