/*
 * mi3d.h - C ABI of libmi3d.so, the MI355X (gfx950) implementation of Make-It-3D's coarse-stage
 * SDS hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 * parameter name ends in `_host`.  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Every entry point enqueues work on `stream` and returns the hipError_t of the launch
 * as int (0 = hipSuccess); nothing here allocates, frees or synchronises.
 *
 * Part 1 replaces, one for one, the 13 functions of the reference's `_raymarching` backend
 * (/root/reference/raymarching/src/raymarching.h:7-22, bound in bindings.cpp:5-23): same argument
 * order and meaning, `at::Tensor` -> raw pointer, plus the trailing stream.  Caller allocates all
 * outputs (as raymarching/raymarching.py does).
 *
 * Part 2 replaces the tiny-cuda-nn `Encoding` (HashGrid) forward/backward the reference calls at
 * /root/reference/nerf/network_tcnn.py:54-65,107.
 *
 * Part 3 evaluates / back-propagates the encoder for the whole 13-point stencil the reference visits with
 * 13 separate encoder passes per sample (network_tcnn.py:115-128, nerf/renderer.py:521-524).
 *
 * Part 4 replaces the three nn.Linear + two ReLU launches of the reference's `sigma_net`
 * (network_tcnn.py:13-32,67, called at :107) by one matrix-core kernel per direction.
 */
#ifndef MI3D_H
#define MI3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI3D_MAX_LEVELS 16
#define MI3D_MAX_POINTS 16

/* the ABI version: 5 (4 = 3 + the compact-round inference loop of Part 1b; 5: that loop's ctl block is int32[16] with
 * the dropped-row count in [8], and its plan never passes max_steps; every other entry point is unchanged) */
int mi3d_abi_version(void);
const char *mi3d_last_error_string(int err);

/* ------------------------------------------------------------------ Part 1: raymarching backend */

/* raymarching.h:7  near_far_from_aabb -- rays_o,rays_d [N,3]; aabb [6]; nears,fars [N] */
int mi3d_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                            float min_near, float *nears, float *fars, void *stream);
/* raymarching.h:8  sph_from_ray -- coords [N,2] */
int mi3d_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                      void *stream);
/* raymarching.h:9  morton3D -- coords int32 [N,3] -> indices int32 [N] */
int mi3d_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream);
/* raymarching.h:10 morton3D_invert */
int mi3d_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream);
/* raymarching.h:11 packbits -- grid float [N*8] -> bitfield uint8 [N] (bit i = grid[8n+i] > thresh) */
int mi3d_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream);

/* raymarching.h:13 march_rays_train -- xyzs,dirs [M,3]; deltas [M,2]; rays int32 [N,3] = (ray id, offset,
 * count); counter int32 [2] (+= samples, += rays).  Rows are written only for rays whose slab fits in M.
 * Slabs are handed out by one atomic per workgroup after an in-workgroup scan (rays[] rows are in ray
 * order: rays[n] describes ray n). */
int mi3d_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                          int32_t *rays, int32_t *counter, const float *noises, void *stream);
/* Not in the reference: zero rows [counter[0], counter[0]+pad) of xyzs/dirs/deltas (pad < align), the rows
 * raymarching.py:237-241 exposes past the last sample; lets the caller skip the 537 MB zero fill of
 * raymarching.py:217-219. */
int mi3d_march_zero_tail(const int32_t *counter, uint32_t align, uint32_t M, float *xyzs, float *dirs,
                         float *deltas, void *stream);

/* raymarching.h:14-15 composite_rays_train_{forward,backward} */
int mi3d_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                      const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                      float *weights_sum, float *depth, float *image, void *stream);
int mi3d_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                       const float *sigmas, const float *rgbs, const float *deltas,
                                       const int32_t *rays, const float *weights_sum, const float *image,
                                       uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                       float *grad_rgbs, void *stream);
/* raymarching.h:16-17 composite_sdf_rays_train_{forward,backward} (alpha = sigma) */
int mi3d_composite_sdf_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                          const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                          float *weights_sum, float *depth, float *image, void *stream);
int mi3d_composite_sdf_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                           const float *sigmas, const float *rgbs, const float *deltas,
                                           const int32_t *rays, const float *weights_sum, const float *image,
                                           uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                           float *grad_rgbs, void *stream);

/* raymarching.h:20 march_rays (inference) */
int mi3d_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                    const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars,
                    float *xyzs, float *dirs, float *deltas, const float *noises, void *stream);
/* raymarching.h:21 composite_rays (inference, in place) */
int mi3d_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                        const float *sigmas, const float *rgbs, const float *normals, const float *deltas,
                        float *weights_sum, float *depth, float *image, float *normal, void *stream);
/* raymarching.h:22 composite_sdf_rays */
int mi3d_composite_sdf_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                            float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                            float *weights_sum, float *depth, float *image, void *stream);

/* ---- Part 1b: the inference loop driven from the device (replaces the host logic of nerf/renderer.py:526-551).
 * The reference loop reads the number of alive rays back every round (a boolean-mask copy = a synchronisation)
 * because launch sizes and n_step = max(min(N / n_alive, 8), 1) depend on it.  Here that state lives in `ctl`
 * (device int32[8]: [0] n_alive, [1] n_step, [2] rows = n_alive*n_step rounded up PAST a multiple of `align` as
 * raymarching.py:397-400 does, [3] marching steps done, [4] rounds done) and the kernels of a round read it; the host
 * launches each round for an upper bound `n_alive_max` of the alive count and may look at ctl[0] as rarely as it likes.
 *   mi3d_infer_begin        rays_alive = 0..N-1, ctl planned for round 0
 *   mi3d_march_rays_ctl     = mi3d_march_rays for ctl's n_alive / n_step; rows a ray leaves unused and the alignment
 *                             rows are zeroed (buffers of N + align rows are re-used across rounds); `noises` (float[N]
 *                             or NULL) jitters round 0 only (renderer.py:546)
 *   mi3d_composite_rays_ctl = mi3d_composite_rays for ctl's n_alive / n_step
 *   mi3d_compact_alive_ctl  rays_alive_out = the entries >= 0 of rays_alive_in[0 .. n_alive), order kept (the boolean
 *                             mask of renderer.py:550); ctl advanced: steps += n_step, n_alive = 0 once steps >=
 *                             max_steps (`while step < max_steps`), next round planned */
int mi3d_infer_begin(int32_t *ctl, int32_t *rays_alive, uint32_t N, uint32_t align, void *stream);
int mi3d_march_rays_ctl(const int32_t *ctl, uint32_t n_alive_max, const int32_t *rays_alive, const float *rays_t,
                        const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t *grid, const float *fars, float *xyzs, float *dirs,
                        float *deltas, const float *noises, void *stream);
int mi3d_composite_rays_ctl(const int32_t *ctl, uint32_t n_alive_max, float T_thresh, int32_t *rays_alive,
                            float *rays_t, const float *sigmas, const float *rgbs, const float *normals,
                            const float *deltas, float *weights_sum, float *depth, float *image, float *normal,
                            void *stream);
int mi3d_compact_alive_ctl(int32_t *ctl, const int32_t *rays_alive_in, int32_t *rays_alive_out, uint32_t N,
                           uint32_t align, uint32_t max_steps, void *stream);

/* The same loop with the samples of a round COMPACTED and the round size set by a row BUDGET (round 5).  The reference's
 * round takes n_step = clamp(N / n_alive, 1, 8) steps of every alive ray and lays them out at n * n_step, so a 128 x 128
 * render is ~280 rounds of at most 16 384 rows - latency, not work.  A ray's result does not depend on how its samples
 * are cut into rounds (it carries t in rays_t and its transmittance in weights_sum), so here a round takes
 * n_step = clamp(budget / n_alive, step_min, step_max) steps and the march packs what the rays actually emitted into one
 * slab per ray (wave scan + one atomic per wave on ctl[2], as the training march does): a render is a handful of rounds
 * and no row of a finished ray is evaluated.  ctl (int32[16], ABI version 5; version 4: int32[8]): [0] n_alive, [1] n_step,
 * [2] rows of this round (written by the march, zeroed by begin / compact), [3] steps done, [4] rounds done, [5] budget,
 * [6] step_min, [7] step_max, [8] rows DROPPED so far because a ray's slab would have passed rows_cap (sticky; zeroed by
 * begin2; non-zero = the render is incomplete: the caller's rows_cap is below max(budget, N step_min)), [9..15] reserved.
 * The plan never lets steps done + n_step pass max_steps (a ray takes at most max_steps steps whatever the budget).
 *   mi3d_infer_begin2            rays_alive[i] = i, ctl for round 0
 *   mi3d_march_rays_compact_ctl  ray_slab int32[n_alive_max][2] = (first row, rows) per alive slot; t_next f32[n_alive_max] =
 *                                the march's own t behind its last step (what the next round resumes from: the restart is
 *                                exact, so the sample sequence of a ray is the same for every budget); a ray whose slab
 *                                would pass rows_cap emits nothing and its rows are counted in ctl[8] (cannot happen when
 *                                rows_cap >= max(budget, N step_min))
 *   mi3d_composite_rays_compact_ctl  composite_rays over each ray's slab; a ray that used all n_step rows resumes at t_next
 *   mi3d_compact_alive_ctl2      the compaction + the next round's plan                                       */
int mi3d_infer_begin2(int32_t *ctl, int32_t *rays_alive, uint32_t N, uint32_t budget_rows, uint32_t step_min,
                      uint32_t step_max, void *stream);
int mi3d_march_rays_compact_ctl(int32_t *ctl, uint32_t n_alive_max, const int32_t *rays_alive, const float *rays_t,
                                const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                uint32_t C, uint32_t H, const uint8_t *grid, const float *fars, uint32_t rows_cap,
                                float *xyzs, float *dirs, float *deltas, int32_t *ray_slab, float *t_next,
                                const float *noises, void *stream);
int mi3d_composite_rays_compact_ctl(const int32_t *ctl, uint32_t n_alive_max, float T_thresh, int32_t *rays_alive,
                                    float *rays_t, const int32_t *ray_slab, const float *t_next, const float *sigmas,
                                    const float *rgbs, const float *normals, const float *deltas, float *weights_sum,
                                    float *depth, float *image, float *normal, void *stream);
int mi3d_compact_alive_ctl2(int32_t *ctl, const int32_t *rays_alive_in, int32_t *rays_alive_out, uint32_t N,
                            uint32_t max_steps, void *stream);

/* ------------------------------------------------------------------ Part 2: hash-grid encoding */

/* Level table of a tcnn HashGrid (host side; no device work).  offsets_host has n_levels+1 entries in
 * units of grid entries (x n_features=2 floats).  Returns the total number of entries. */
uint32_t mi3d_hashgrid_levels(uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                              uint32_t log2_hashmap_size, uint32_t *offsets_host, uint32_t *resolutions_host,
                              float *scales_host);

/* tcnn.Encoding.forward: x [n,3] in [0,1]; params fp32 [entries*2]; out [n, n_levels*2] (feature = level*2+f) */
int mi3d_hashgrid_forward(const float *x, uint32_t n, const float *params, uint32_t n_levels,
                          uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size,
                          float *out, void *stream);
/* tcnn.Encoding.backward wrt params: grad_params [entries*2] is ACCUMULATED into (caller zeroes it) */
int mi3d_hashgrid_backward(const float *x, uint32_t n, const float *dout, uint32_t n_levels,
                           uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size,
                           float *grad_params, void *stream);

/* ------------------------------------------------------------------ Part 3: stencil-aware grid ops */

/* The reference evaluates the field at 13 points per sample: x, x +- eps e_i (finite_difference_normal,
 * network_tcnn.py:115-128) and the same six offsets around x2 = x + 0.01 randn (loss_smooth,
 * nerf/renderer.py:521-524), each as a separate encoder pass.  These two entry points take the whole stencil:
 * point p of sample i is clamp(base + offsets[p], -bound, bound) with base = x[i] for p < P0 and x2[i] for
 * P0 <= p < P (x2 may be NULL when P0 == P), mapped to [0,1] as (pt + bound) / (2 bound) (network_tcnn.py:106).
 * Rows are POINT-MAJOR: row (p*n + i) of `out` / `dout` holds the n_levels*2 features of point p of sample i, so the
 * rows of one stencil point are contiguous (a backward pass that only reaches the first P' points - sigma / albedo
 * alone reach point 0, the normal points 0..6 - runs on that prefix of the rows and nothing else).
 * `count` (device int32, may be NULL) caps n at min(n, *count) without a host round trip.
 * `step` (scatter only): the marching step dt_min in world units (0 = unknown); it only selects which levels
 * use run-merging vs lane-quad atomics, never the result. */
int mi3d_grid_encode_points(const float *x, const float *x2, uint32_t n, const int32_t *count,
                            const float *offsets_host, uint32_t P0, uint32_t P, float bound, const float *params,
                            uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                            uint32_t log2_hashmap_size, float *out, void *stream);
/* mi3d_grid_encode_points with level-major output planes [n_levels][P*n][2] (feature pair of level l, row r = p*n + i
 * at out_planes[(l*P*n + r)*2]) - the layout the MLP kernels take with x_plane_rows = P*n.  The (level, tile) work is
 * tied to XCDs so each XCD's L2 only ever holds the table of the level it is gathering from; `step` (the marching
 * step in world units, 0 = unknown) only balances that split, never the result.  The workgroups of an XCD claim their
 * tiles from per-segment counters: 512 bytes of a 64-slot ring in device memory that belongs to the library, zeroed
 * in `stream` before the launch.  Calls on ONE stream may be queued without limit (a slot coming round again is behind
 * its previous user in stream order); a call that finds its slot last used by ANOTHER stream that has not drained deals
 * its tiles statically instead of claiming them (same planes, slower) - two live launches never share counters.
 * PLANE ELEMENT TYPE: out_half == 0: fp32 pairs (8 bytes per (level, row)); out_half != 0: binary16 pairs (4 bytes) -
 * for use under torch.autocast(float16) only, where the first nn.Linear rounds its input to binary16 anyway
 * (mi3d_mlp_forward / _backward with half_mode != 0 read them with planes_half != 0: same MLP output, bit for bit,
 * half the bytes).  The gradient planes mi3d_mlp_backward writes follow the same flag: the input gradient of that
 * Linear comes out of a binary16 GEMM in the reference, so binary16 planes hold exactly what autocast would hand the
 * encoder's backward. */
int mi3d_grid_encode_points_planes(const float *x, const float *x2, uint32_t n, const float *offsets_host, uint32_t P0,
                                   uint32_t P, float bound, const float *params, uint32_t n_levels,
                                   uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size, float step,
                                   void *out_planes, int out_half, void *stream);
/* ... with a DEVICE-side sample count (count == NULL: all n): samples s >= *count are not evaluated, their rows are not
 * written; the plane strides stay n.  Used by the inference loop, whose row count lives in its control block. */
int mi3d_grid_encode_points_planes_counted(const float *x, const float *x2, uint32_t n, const int32_t *count,
                                           const float *offsets_host, uint32_t P0, uint32_t P, float bound,
                                           const float *params, uint32_t n_levels, uint32_t base_resolution,
                                           float per_level_scale, uint32_t log2_hashmap_size, float step, void *out_planes,
                                           int out_half, void *stream);
int mi3d_grid_scatter_points(const float *x, const float *x2, uint32_t n, const int32_t *count,
                             const float *offsets_host, uint32_t P0, uint32_t P, float bound, const float *dout,
                             uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                             uint32_t log2_hashmap_size, float step, float *grad_params, void *stream);

/* The same scatter without global atomics: every corner contribution (on the coarse levels: what a wave's 64 samples
 * x P points contribute to one entry, summed in LDS first; on the fine levels: the two x-neighbours of a corner pair as
 * ONE 16-byte record) is appended to the region of its 64-KB gradient bin, then each bin is accumulated in LDS in 64-bit fixed point and
 * added to the table (see hashgrid.hip).  fp32 contributions throughout; a non-finite contribution bypasses the
 * records and is added to the table with float atomics, so the inf / NaN lands on exactly the entries the reference's
 * atomicAdd would have put it on (and nowhere else).
 * `dout_planes` is level-major [n_levels][P*n][2], rows point-major (what mi3d_mlp_backward writes with
 * dx_plane_rows = P*n); dout_half != 0: binary16 pairs (see mi3d_grid_encode_points_planes).
 * `workspace` is caller-provided device scratch (never allocated here); samples are processed in the FEWEST equal
 * slices whose record arena fits it (ceil(n / k), k = 1, 2, 3 ...; the emit's tile-claim counters live in it too); with workspace == NULL or too small for even 64 samples the atomic kernels of mi3d_grid_scatter_points run
 * instead.  mi3d_grid_scatter_binned_workspace() returns the size that lets n samples go in ONE slice. */
size_t mi3d_grid_scatter_binned_workspace(uint32_t n, uint32_t P, float bound, float step, uint32_t n_levels,
                                          uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size);
int mi3d_grid_scatter_binned(const float *x, const float *x2, uint32_t n, const float *offsets_host, uint32_t P0,
                             uint32_t P, float bound, const void *dout_planes, int dout_half, uint32_t n_levels,
                             uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size, float step,
                             void *workspace, size_t workspace_bytes, float *grad_params, void *stream);
/* ... plus a SECOND gradient pair for stencil point 0: `extra_point0_planes` [n_levels][n][2] (same element type as
 * dout_planes; NULL = none).  The reference back-propagates twice through one forward (nerf/sd.py:171 latents.backward,
 * then nerf/utils.py:983 scaler.scale(loss).backward()); the first pass reaches sigma / albedo of point 0 only.  Its
 * point-0 gradient planes can be handed to the second pass's scatter here, which adds the two pairs of point 0 in fp32
 * and scatters the sum - the same table gradient as two scatters (tcnn's atomics add the two products separately: equal
 * up to fp32 rounding of w (a + b) against w a + w b), for one pass over the table instead of two. */
int mi3d_grid_scatter_binned_plus(const float *x, const float *x2, uint32_t n, const float *offsets_host, uint32_t P0,
                                  uint32_t P, float bound, const void *dout_planes, const void *extra_point0_planes,
                                  int dout_half, uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                                  uint32_t log2_hashmap_size, float step, void *workspace, size_t workspace_bytes,
                                  float *grad_params, void *stream);

/* Host-side planning queries: no device work, callable without a GPU (tests/test_plan_cpu.py).
 * mi3d_grid_encode_plan: how mi3d_grid_encode_points_planes cuts the (level, tile-of-64-samples) list into one run of
 *   segments per XCD: n_segments[8], segments[8][16][3] = (level, first tile, end tile).
 * mi3d_grid_scatter_plan: how mi3d_grid_scatter_binned lays out its workspace for n samples and `workspace_bytes`:
 *   out[0] samples per slice, out[1] workspace bytes one slice uses, out[2] coarse levels (gathered per tile),
 *   out[3] reduce workgroups, out[4] record-arena bytes, out[5] region counters, then 7 values per level: bins, region
 *   capacity (records), emitting waves, 1 = 16-byte x-pair records, reduce workgroups per bin, first reduce workgroup,
 *   first region counter.  out must hold 6 + 7 * n_levels values.  (The layout for fp32 gradient planes, 16-byte pair
 *   records - what mi3d_grid_scatter_binned_workspace sizes; with binary16 planes the pair records take 12 bytes, so the
 *   same workspace holds more samples per slice than this query says.)
 * mi3d_grid_level_routes: which index route the plane gather and the scatter's emit take on each level (kinds[n_levels]):
 *   0 the general rule (any dims, any table size, any input - tcnn's grid_index as the oracle restates it), 1 dense 3-D
 *   strided (24-bit multiplies, at most one wrap), 2 power-of-two hash (mask).  `stencil_points` != 0: the positions are
 *   clamped stencil points (mi3d_grid_encode_points*), 0: raw positions (mi3d_hashgrid_*), which always take route 0. */
int mi3d_grid_encode_plan(uint32_t n, float bound, float step, uint32_t n_levels, uint32_t base_resolution,
                          float per_level_scale, uint32_t log2_hashmap_size, uint32_t *n_segments, uint32_t *segments);
int mi3d_grid_scatter_plan(uint32_t n, uint32_t P, float bound, float step, uint32_t n_levels, uint32_t base_resolution,
                           float per_level_scale, uint32_t log2_hashmap_size, size_t workspace_bytes,
                           unsigned long long *out);
int mi3d_grid_level_routes(uint32_t n_levels, uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size,
                           int stencil_points, int32_t *kinds);

/* ------------------------------------------------------------------ Part 4: the field's MLP (sigma_net) */

/* network_tcnn.py:13-32: y = W3 relu(W2 relu(W1 x + b1) + b2) + b3, torch nn.Linear layout (W_l is [out_l, in_l]
 * row-major fp32, the master weights).  x [n, dim_in] fp32 rows (x_plane_rows == 0), or level-major planes
 * [dim_in/2][x_plane_rows][2] of which the first n rows are processed (x_plane_rows >= n; planes_half != 0: the planes
 * hold binary16 pairs, half_mode only - applies to x and, in the backward, to dx alike); out [n, dim_out] fp32.
 * half_mode != 0 reproduces torch.autocast(float16) around the stack (nerf/utils.py:979): inputs, weights, biases
 * and every layer output are rounded to binary16, products accumulate in fp32 (v_mfma_f32_32x32x16_f16);
 * half_mode == 0 is exact fp32 (v_mfma_f32_32x32x2_f32).  Supported shapes - every one the reference's MLP class can
 * build around this field (network_tcnn.py:37-45,67 takes num_layers and hidden_dim; BASELINE config 1 is 8 -> 32 -> 4):
 * dim_in even, 2..32 (= 2 x grid levels), dim_hidden 32 or 64, dim_out 4, two or three layers.  TWO layers are
 * requested by passing W2 == b2 == NULL (and dW2 == db2 == NULL in the backward): y = W3 relu(W1 x + b1) + b3.
 * Anything else returns hipErrorInvalidValue - ask mi3d_mlp_supported() first. */
int mi3d_mlp_supported(uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out, uint32_t num_layers);
int mi3d_mlp_forward(const void *x, uint32_t x_plane_rows, int planes_half, uint32_t n, const float *W1,
                     const float *b1, const float *W2, const float *b2,
                     const float *W3, const float *b3, uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out,
                     int half_mode, float *out, void *stream);
/* Backward of the above for upstream gradient dout [n, dim_out]: writes dx and ACCUMULATES the weight and bias
 * gradients (fp32, same layouts as the weights; caller zeroes them).  Activations are recomputed.
 * dx_plane_rows == 0: dx is [n, dim_in] rows; otherwise dx is level-major planes [dim_in/2][dx_plane_rows][2]
 * (feature pair (2l, 2l+1) of row r at dx[(l*dx_plane_rows + r)*2]), the layout mi3d_grid_scatter_binned consumes. */
/* The same with a DEVICE-side row count (the inference loop's control block, Part 1b): rows are point-major with
 * `n_stride` rows per stencil point, and only samples s < *count carry data; tiles wholly beyond it are skipped and
 * their outputs left untouched.  count == NULL: every row.  (The gather and the head have the same variant.) */
int mi3d_mlp_forward_counted(const void *x, uint32_t x_plane_rows, int planes_half, uint32_t n, const int32_t *count,
                             uint32_t n_stride, const float *W1, const float *b1, const float *W2, const float *b2,
                             const float *W3, const float *b3, uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out,
                             int half_mode, float *out, void *stream);
int mi3d_mlp_backward(const void *x, uint32_t x_plane_rows, int planes_half, const float *dout, uint32_t n,
                      const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                      const float *b3, uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out, int half_mode, void *dx,
                      uint32_t dx_plane_rows, float *dW1, float *db1, float *dW2, float *db2, float *dW3, float *db3,
                      void *stream);

/* ------------------------------------------------------------------ Part 5: the field head */

/* From the MLP output h [P, n, 4] (point-major rows p*n + i) of the stencil points (P = 7: sample + its six +-epsilon neighbours in the order
 * +x,-x,+y,-y,+z,-z; P = 13: plus the six neighbours of x2) to what the renderer consumes, in one elementwise pass:
 *   sigma [n]     = exp(h_0[0] + blob(x))                         network_tcnn.py:94-100,109, activation.py:5-18
 *   albedo [n,3]  = sigmoid(h_0[1..3])                            network_tcnn.py:110
 *   normal [n,3]  = nan_to_num(safe_normalize(-(sigma+ - sigma-) / (2 epsilon)))   network_tcnn.py:115-138, utils.py:47-48
 *   normal2 [n,3] = the same around x2 (P = 13 only)              nerf/renderer.py:521-524
 * The stencil positions are clamp(base + offsets_host[p], -bound, bound), as in mi3d_grid_encode_points. */
int mi3d_field_head_forward(const float *h, const float *x, const float *x2, uint32_t n, const float *offsets_host,
                            uint32_t P, float bound, float blob_density, float blob_radius, float epsilon, float *sigma,
                            float *albedo, float *normal, float *normal2, void *stream);
int mi3d_field_head_forward_counted(const float *h, const float *x, const float *x2, uint32_t n, const int32_t *count,
                                    const float *offsets_host, uint32_t P, float bound, float blob_density,
                                    float blob_radius, float epsilon, float *sigma, float *albedo, float *normal,
                                    float *normal2, void *stream);
/* Backward: upstream gradients (any may be NULL = zero) -> dh [P_active, n, 4], the rows of the first P_active
 * points only: 1 (sigma / albedo alone carry a gradient: the SDS pass), 7 (+ normal) or 13 (+ normal2); the caller
 * runs the MLP backward and the scatter over that prefix.  trunc_exp's clamped derivative (activation.py:15-18),
 * clamp and nan_to_num pass gradients exactly where torch's do. */
int mi3d_field_head_backward(const float *h, const float *x, const float *x2, uint32_t n, const float *offsets_host,
                             uint32_t P, uint32_t P_active, float bound, float blob_density, float blob_radius,
                             float epsilon, const float *dsigma, const float *dalbedo, const float *dnormal,
                             const float *dnormal2, float *dh, void *stream);

/* ------------------------------------------------------------------ Part 6: the optimizer step */

/* Adan as the reference configures it (main.py:132; optimizer.py:100-249 `Adan.step` + `_single_tensor_adan`), one
 * fused elementwise pass per parameter tensor, in the reference's operation order.  All pointers are device fp32
 * arrays of `count` elements, 16-byte aligned; grad is scaled by the clip factor in place (as the reference does),
 * the four state arrays are updated in place (the caller zero-initialises exp_avg / exp_avg_sq / exp_avg_diff once;
 * neg_pre_grad is initialised here when first_step != 0).
 * The global-norm clip stays on the device: call mi3d_sumsq_accumulate on every gradient tensor into ONE zeroed device
 * float, pass it as grad_sumsq; clip = min(max_grad_norm / (sqrt(sum) + clip_eps), 1) (optimizer.py:107-127, without
 * its `.item()` host sync).  grad_sumsq == NULL or max_grad_norm <= 0: no clipping. */
int mi3d_sumsq_accumulate(const float *x, size_t n, float *acc, void *stream);
int mi3d_adan_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *exp_avg_diff,
                   float *neg_pre_grad, size_t count, const float *grad_sumsq, float max_grad_norm, float clip_eps,
                   int first_step, float beta1, float beta2, float beta3, float bias_correction1,
                   float bias_correction2, float bias_correction3_sqrt, float lr, float weight_decay, float eps,
                   int no_prox, void *stream);

/* ------------------------------------------------------------------ Part 7: refine-stage point renderer */

/* The two pytorch3d calls of the reference's `render_point` (nerf/refine_utils.py:306-333).
 * mi3d_points_rasterize = pytorch3d.renderer.points.rasterize_points on ONE point cloud: points_ndc [P,3] = (x, y in
 * pytorch3d NDC: +X left, +Y up; z = depth, points with z < 0 are skipped), image H x W, `radius` in NDC units,
 * K = points_per_pixel <= 8.  Outputs [H,W,K]: idx (point index, -1 = unused), zbuf (may be NULL), dists (squared NDC
 * distance to the pixel centre, -1 = unused), the K nearest covering points in ascending z (ties: lower index first).
 * `workspace` holds the per-tile point lists (mi3d_points_rasterize_workspace bytes cover the worst case).
 * mi3d_points_composite_forward = alphas = 1 - sqrt(clamp(0.1 dists / radius^2, 1e-3, 1)) (refine_utils.py:321-326)
 * followed by compositing.alpha_composite: out [C,H,W] from features [P,C] (C <= 32), front to back.
 * mi3d_points_composite_backward ACCUMULATES d out / d features into grad_features [P,C] (caller zeroes it); the point
 * positions carry no gradient on this path (the reference optimises colours and features, nerf/utils.py:826-831). */
size_t mi3d_points_rasterize_workspace(uint32_t P, uint32_t H, uint32_t W, float radius);
int mi3d_points_rasterize(const float *points_ndc, uint32_t P, uint32_t H, uint32_t W, float radius,
                          uint32_t points_per_pixel, void *workspace, size_t workspace_bytes, int32_t *idx, float *zbuf,
                          float *dists, void *stream);
int mi3d_points_composite_forward(const int32_t *idx, const float *dists, uint32_t H, uint32_t W, uint32_t points_per_pixel,
                                  const float *features, uint32_t C, double radius, float *out, void *stream);
int mi3d_points_composite_backward(const int32_t *idx, const float *dists, uint32_t H, uint32_t W,
                                   uint32_t points_per_pixel, const float *grad_out, uint32_t C, double radius,
                                   float *grad_features, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI3D_H */
