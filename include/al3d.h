/*
 * al3d.h -- C ABI of libal3d_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hot path of the diversity-based active-learning
 * selector: every pointer is a DEVICE pointer unless stated otherwise, every
 * size is explicit, outputs and workspaces are caller-allocated, `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Functions return 0 on
 * success or a negative AL3D_E* code; al3d_last_error() returns a thread-local
 * message.  No exceptions, no global state besides the error string, no
 * allocation and no synchronisation inside any entry point (graph-capturable).
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the reference repository root).
 */
#ifndef AL3D_H_
#define AL3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AL3D_OK 0
#define AL3D_EINVAL (-1)   /* bad argument (shape, alignment, enum) */
#define AL3D_ELAUNCH (-2)  /* HIP launch failure */
#define AL3D_ENOSPC (-3)   /* caller-provided capacity too small */

/* normalize / aggregate enums of al3d_combine_maps_f64 */
#define AL3D_NORM_NONE 0
#define AL3D_NORM_EXP 1     /* 1 - exp(-x) */
#define AL3D_NORM_LINEAR 2  /* x / scale   */
#define AL3D_AGG_SUM 0
#define AL3D_AGG_MIN 1
#define AL3D_AGG_MAX 2

/* greedy status words written to out_meta[1] */
#define AL3D_GREEDY_OK 0
#define AL3D_GREEDY_DUPLICATE (-1) /* reference `assert selected_index not in ...` would fire */
#define AL3D_GREEDY_FULL (-2)      /* out_idx capacity exhausted */

int al3d_abi_version(void);
const char* al3d_last_error(void);

/* ---------------------------------------------------------------- selector */

/* Pairwise embedding distance map, float32 [n,n] from feats [n,c].
 * Replaces FeatureSelector.get_feature_distance_map
 * (det3d/selectors/feature_selector.py:87-109; same body in
 * spatial_temporal_feature_selector.py:97-111): p==1 -> sum |a-b|,
 * p==2 -> sum sqrt((a-b)^2)  (also L1, bug-compatible).  Sum order c=0..C-1. */
int al3d_l1_distance_f32(const float* feats, int64_t n, int64_t c, int p,
                         float* out, void* stream);
/* Rows [row0, row0+nrows) of the same map into out [nrows, n] (16-byte aligned): the N>1 path
 * computes one block of rows per rank and all-gathers them. */
int al3d_l1_distance_rows_f32(const float* feats, int64_t n, int64_t c, int p, int64_t row0,
                              int64_t nrows, float* out, void* stream);

/* Normalise + aggregate the spatial (f64 [n,n] or NULL), temporal (derived on
 * the fly from temporal_id [n] or NULL: |i-j| if ids equal else 1e6) and
 * feature (f32 [n,n] or NULL) terms into out f64 [n,n].
 * Replaces det3d/selectors/spatial_temporal_selector.py:109-155,
 * spatial_temporal_feature_selector.py:187-219, spatial_feature_selector.py:188-197,
 * temporal_selector.py:56-63. */
int al3d_combine_maps_f64(const double* spatial, const int64_t* temporal_id, const float* feat,
                          int64_t n, int normalize, int aggregate,
                          double lambda_t, double lambda_f,
                          double spatial_scale, double temporal_scale,
                          double* out, void* stream);

/* Same-location Euclidean map, f64 [n,n]: sqrt(dx^2+dy^2) if loc_id equal else 1e6.
 * Replaces det3d/selectors/euclidean_spatial_selector.py:95-106. */
int al3d_euclid_map_f64(const double* xy, const int64_t* loc_id, int64_t n, double* out,
                        void* stream);

/* max over finite entries of a[0..count) -> *out_dev (device f64; -inf if none).
 * Replaces `spatial_distance_map[spatial_distance_map != np.inf].max()`
 * (spatial_temporal_selector.py:139). */
int al3d_max_finite_f64(const double* a, int64_t count, double* out_dev, void* stream);

/* Greedy k-center (farthest-point) selection under a cost budget, one
 * persistent workgroup.  Replaces the loop at
 * det3d/selectors/spatial_temporal_selector.py:157-193 (float64 numpy) and
 * det3d/selectors/feature_selector.py:142-172 (float32 torch).
 *   D, seed_map   [n,n] row-major; seed_map initialises fps (== D except
 *                 SpatialFeatureSelector)
 *   seeded        [n_seeded] already-sampled ids; if n_seeded == 0, `first` is
 *                 the initial pick (python random.choice on the host)
 *   box_cost      [n] f64 = n_boxes[i] * cost_b
 *   out_idx       [cap] picks in order;  out_meta[0] = count, out_meta[1] = status
 *   workspace     >= al3d_greedy_workspace_bytes(n, elem_size) bytes */
int64_t al3d_greedy_workspace_bytes(int64_t n, int elem_size);
int al3d_greedy_kcenter_f64(const double* D, const double* seed_map, int64_t n,
                            const int64_t* seeded, int64_t n_seeded, int64_t first,
                            const double* box_cost, double cost_f, double start_cost,
                            double budget_int, int check_seeded,
                            int64_t* out_idx, int64_t cap, int64_t* out_meta,
                            void* workspace, void* stream);
int al3d_greedy_kcenter_f32(const float* D, const float* seed_map, int64_t n,
                            const int64_t* seeded, int64_t n_seeded, int64_t first,
                            const double* box_cost, double cost_f, double start_cost,
                            double budget_int, int check_seeded,
                            int64_t* out_idx, int64_t cap, int64_t* out_meta,
                            void* workspace, void* stream);

/* Uncertainty-selector epilogues.
 * frame entropy: mean over a frame's kept boxes of -s log s - (1-s) log(1-s), read from the
 * al3d_head_decode_nms outputs (scores [B,nt,post], counts [B,nt]); empty frame -> NaN.
 * Replaces det3d/selectors/entropy_selector.py:72-75 (same lines in badge/uwe selectors). */
int al3d_frame_entropy_f32(const float* scores, const int* counts, int B, int nt, int post,
                           float* out, void* stream);
/* PPAL: sum over a frame's kept boxes of entropy * class_weight[label] (labels [B,nt,post] i32,
 * class_weight [ncls] f32); empty frame -> 0.  Replaces det3d/selectors/ppal_selector.py:99-109. */
int al3d_frame_weighted_entropy_f32(const float* scores, const int* labels, const int* counts, int B,
                                    int nt, int post, const float* class_weight, int ncls, float* out,
                                    void* stream);
/* PPAL pool restriction: D[i,j] = -inf unless keep[i] && keep[j] (ppal_selector.py:194-196) */
int al3d_mask_map_f32(float* D, int64_t n, const unsigned char* keep, void* stream);
/* out[i,:] = feats[i,:] * w[widx ? widx[i] : i]  (badge_selector.py:75-78, uwe_selector.py:96-99) */
int al3d_scale_rows_f32(const float* feats, const float* w, const int64_t* widx, int64_t n, int c,
                        float* out, void* stream);
/* (x - min) / (max - min), NaN-propagating like torch (uwe_selector.py:78) */
int al3d_minmax_norm_f32(const float* x, int64_t n, float* out, void* stream);
/* torch.argsort(-x): descending, NaN last, ties by ascending index (entropy_selector.py:121) */
int64_t al3d_argsort_workspace_bytes(int64_t n);
int al3d_argsort_desc_f32(const float* x, int64_t n, int64_t* out_idx, void* workspace, void* stream);

/* Exact k-nearest neighbours (self included) of 2-D points, ascending
 * (distance, index).  Replaces scipy cKDTree(locations).query(locations, k+1)
 * (spatial_temporal_selector.py:97-98).  xy [n,2] f64 -> knn_d [n,kq] f64,
 * knn_i [n,kq] i64 (padded with inf / n when n < kq). kq <= 32. */
int al3d_knn_2d_f64(const double* xy, int64_t n, int kq, double* knn_d, int64_t* knn_i,
                    void* stream);

/* All-pairs shortest paths over the symmetrised kNN graph (zero-length edges
 * dropped), f64 [n,n], unreachable = +inf.  Bit-identical to Dijkstra's
 * left-to-right path sums.  Replaces the dense edge-matrix build +
 * scipy.sparse.csgraph.shortest_path(directed=False, method="D")
 * (spatial_temporal_selector.py:95-104). */
int64_t al3d_apsp_workspace_bytes(int64_t n, int kq);
int al3d_apsp_knn_f64(const double* knn_d, const int64_t* knn_i, int64_t n, int kq,
                      double* out, void* workspace, void* stream);
/* Same, source rows [row0, row0+nrows) only -> out [nrows, n] (row-sharding across ranks). */
int al3d_apsp_knn_rows_f64(const double* knn_d, const int64_t* knn_i, int64_t n, int kq,
                           int64_t row0, int64_t nrows, double* out, void* workspace, void* stream);

/* ---------------------------------------------------------------- detector: sweep merge (a1) */

/* One frame's point cloud from its key-frame file and sweep files, on device.  Replaces read_file /
 * remove_close / read_sweep / LoadPointCloudFromFile.__call__ (NuScenesDataset branch)
 * (det3d/datasets/pipelines/loading.py:17-63,98-126).
 *   raw        [total_rows, 5] f32 rows (x, y, z, intensity, ring) of all files back to back;
 *              file f owns rows [file_off[f], file_off[f+1]) (file_off: nfiles+1 int64, device)
 *   file 0     is the key frame: kept whole, not moved, time 0
 *   file f>0   drops points with |x| < min_distance && |y| < min_distance (in the sweep's own
 *              frame), then, if has_xform[f], is moved by xform[12*f .. 12*f+12) = the first three
 *              rows of the reference's float64 transform_matrix (row-major 3x4), evaluated as
 *              ((T0*x + T1*y) + T2*z) + T3 in float64 and rounded once to float32; its points get
 *              time = (float)time_lag[f]
 *   out        [<= total_rows, 5] f32 (x, y, z, intensity, dt), file order then point order
 *   out_count  rows written (device int)
 *   workspace  >= al3d_merge_sweeps_workspace_bytes(total_rows) */
int64_t al3d_merge_sweeps_workspace_bytes(int64_t total_rows);
int al3d_merge_sweeps_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                          const double* xform, const unsigned char* has_xform, const double* time_lag,
                          float min_distance, float* out, int* out_count, void* workspace, void* stream);

/* ---------------------------------------------------------------- detector: voxelize */

/* Voxelise a batch of point clouds and reduce each voxel to its mean point (VFE).
 * Replaces Voxelization/VoxelGenerator.generate -> points_to_voxel_new
 * (det3d/datasets/pipelines/preprocess.py:275-304,
 *  det3d/ops/point_cloud/point_cloud_ops.py:213-296) for every frame of the batch,
 * collate_kitti's concatenation with a batch index (det3d/torchie/parallel/collate.py:118-131)
 * and VoxelFeatureExtractorV3.forward (det3d/models/readers/voxel_encoder.py:206-211).
 *   points        [npts, nfeat] f32, frames concatenated; point_offsets [B+1] i64 (device)
 *   range_min / voxel_size / grid_size   HOST arrays of 3 (x, y, z)
 *   first_grid    persistent i32 scratch of al3d_voxelize_grid_bytes() bytes, initialised
 *                 once with al3d_voxelize_grid_init(); left clean by every call
 *   feat          [B*max_voxels, nfeat] f32 mean features (rows >= total are untouched)
 *   coords        [B*max_voxels, 4] i32 (batch, z, y, x), first-appearance order per frame
 *   num_points    [B*max_voxels] i32 clipped counts; voxels [B*max_voxels, max_points, nfeat]
 *                 zero-padded point slots or NULL; num_voxels [B] i32; row_base [B+1] i32
 *                 (row_base[B] = total voxels). */
int64_t al3d_voxelize_grid_bytes(int B, int gx, int gy, int gz);
int al3d_voxelize_grid_init(void* grid, int B, int gx, int gy, int gz, void* stream);
int64_t al3d_voxelize_workspace_bytes(int64_t npts, int B, int max_voxels);
int al3d_voxelize_mean_f32(const float* points, const int64_t* point_offsets, int64_t npts, int B,
                           int nfeat, const float* range_min, const float* voxel_size,
                           const int* grid_size, int max_points, int max_voxels, void* first_grid,
                           void* workspace, float* feat, int* coords, int* num_points, float* voxels,
                           int* num_voxels, int* row_base, void* stream);

/* Mean VFE on reference-format input: voxels [m,max_points,nfeat] zero-padded, counts [m] i32
 * -> feat [m,nfeat].  Replaces VoxelFeatureExtractorV3.forward
 * (det3d/models/readers/voxel_encoder.py:206-211). */
int al3d_vfe_mean_f32(const float* voxels, const int* num_points, int m, int max_points, int nfeat,
                      float* feat, void* stream);

/* ---------------------------------------------------------------- detector: head */

/* Decode + score + top-k + rotated NMS + range mask for every (sample, task) pair, one
 * workgroup each.  hout is the fused NHWC head output [B, HW, CH]; task t reads its box
 * codes at channels [box_off[t], +na*10) and class logits at [cls_off[t], +na*nc).
 * anchors: HOST array of ntasks DEVICE pointers to [A_t, 9]; the int arrays are HOST arrays.
 * Outputs [B, ntasks, post_max, *] + counts [B, ntasks].
 * Replaces MultiGroupHead.predict/get_task_detections (det3d/models/bbox_heads/mg_head.py:697-1085),
 * second_box_decode (det3d/core/bbox/box_torch_ops.py:80-148), rotate_nms
 * (box_torch_ops.py:528-550) and rotate_non_max_suppression_cpu (det3d/ops/nms/nms_cpu.h:73-168). */
int al3d_head_decode_nms(const float* hout, int B, int HW, int CH, int ntasks,
                         const float* const* anchors, const int* task_A, const int* task_na,
                         const int* task_nc, const int* box_off, const int* cls_off,
                         const int* label_off, float score_thresh, float iou_thresh, int pre_max,
                         int post_max, const float* range6, float* boxes, float* scores, int* labels,
                         int* counts, void* workspace, void* stream);
/* workspace of al3d_head_decode_nms: one 32-bit score word per (sample, task, anchor); a whole-GPU
 * pre-pass fills it so that the per-(sample, task) workgroups stream compact scores instead of
 * re-reading the strided head output in every select pass.  task_A: host array [ntasks]. */
int64_t al3d_head_decode_nms_workspace_bytes(int B, int ntasks, const int* task_A);

/* Residual box decode (9-dim boxes, vector-encoded angle): enc [n,10], anchors [n,9] -> [n,9].
 * Replaces GroundBox3dCoderTorch.decode_torch -> second_box_decode
 * (det3d/core/bbox/box_coders.py:106-109, det3d/core/bbox/box_torch_ops.py:80-148). */
int al3d_box_decode_f32(const float* enc, const float* anchors, int64_t n, float* out, void* stream);

/* ---------------------------------------------------------------- detector: sparse encoder */

/* Sparse 3-D convolution pieces.  Replace the spconv calls of FPNSpMiddleResNetFHD
 * (det3d/models/backbones/scn.py:28-97,316-392: SubMConv3d / SparseConv3d /
 * SparseConvTensor.dense()).  coords are [n,4] i32 (batch, z, y, x); `grid` is a dense
 * per-level index grid [B,D,H,W] i32 holding the row id of every active site (-1 = empty)
 * that the caller keeps in HBM across calls.  The rulebook is stored tap-major:
 * nbr[k * n_out + o] = input row feeding output o through kernel offset k = (kz*kh+ky)*kw+kx,
 * or -1 (one entry per (offset, output) pair; no scatter, no atomics in the conv). */
int al3d_sp_fill_i32(int* buf, int64_t count, int value, void* stream);          /* 0 or -1 */
int al3d_sp_scatter_index(const int* coords, int n, int B, int D, int H, int W, int* grid,
                          int mode /* 1: grid=row id, 0: grid=-1 */, void* stream);
int al3d_sp_subm_table(const int* coords, int n, int B, int D, int H, int W, const int* grid,
                       int kd, int kh, int kw, int* nbr, void* stream);
/* strided conv: discover output sites (coords_out rows in arbitrary order, *counter = count,
 * grid_out filled with their row ids) ... ksize/stride/pad are HOST int[3] (z, y, x). */
int al3d_sp_down_claim(const int* coords_in, int n_in, const int* ksize, const int* stride,
                       const int* pad, int B, int OD, int OH, int OW, int* grid_out,
                       int* coords_out, int* counter, int cap, void* stream);
/* Deterministic variant used by the encoder: mark the fed output cells, scan the grid, number
 * the sites in raster (b,z,y,x) order.  Same outputs as al3d_sp_down_claim but with a fixed
 * row order; workspace >= al3d_sp_down_sites_workspace_bytes(). */
int64_t al3d_sp_down_sites_workspace_bytes(int B, int OD, int OH, int OW);
int al3d_sp_down_sites(const int* coords_in, int n_in, const int* ksize, const int* stride,
                       const int* pad, int B, int OD, int OH, int OW, int* grid_out,
                       int* coords_out, int* counter, int cap, void* workspace, void* stream);
/* The same enumeration with the sites numbered COLUMN BY COLUMN: 8 x 8 (y, x) columns over all z, in (b, y/8, x/8,
 * z, y%8, x%8) order -- the order of a level's rows is free inside the encoder (spconv's own order, indice.cu's atomics,
 * is arbitrary), and this one makes R consecutive rows a compact patch whose z neighbours lie inside it, which is what
 * al3d_sp_conv_blk_f16x3 stages.  Same outputs otherwise (grid_out in the plain (b, z, y, x) layout);
 * workspace >= al3d_sp_down_sites_blocked_workspace_bytes().  Reference: spconv_ops.h:51-120 (getIndicePair). */
int64_t al3d_sp_down_sites_blocked_workspace_bytes(int B, int OD, int OH, int OW);
int al3d_sp_down_sites_blocked(const int* coords_in, int n_in, const int* ksize, const int* stride,
                               const int* pad, int B, int OD, int OH, int OW, int* grid_out,
                               int* coords_out, int* counter, int cap, void* workspace, void* stream);
/* Re-numbers the rows of a level (coords in raster order, grid[cell] = row, as al3d_sp_down_sites leaves them) inside windows of
 * `window` (1024 | 4096 | 8192 | 16384) consecutive rows by their own 27-tap submanifold neighbour mask, so that the rows of a
 * 32-row tile tend to lack the SAME taps and the conv kernels' whole-tile tap skip fires more often (the order of a level's rows
 * is free: spconv's own, indice.cu's atomics, is arbitrary).  coords_out [n, 4] = the permuted coords (must not alias coords);
 * grid is renumbered in place; workspace >= al3d_sp_mask_window_sort_workspace_bytes(n).  Call before any table of the level
 * is built.  Reference: spconv_ops.h:51-120 (getIndicePair). */
int64_t al3d_sp_mask_window_sort_workspace_bytes(int n);
int al3d_sp_mask_window_sort(const int* coords, int n, int B, int D, int H, int W, int* grid, int window,
                             int* coords_out, void* workspace, void* stream);
/* ... then its rulebook from the input level's grid */
int al3d_sp_down_table(const int* coords_out, int n_out, const int* ksize, const int* stride,
                       const int* pad, int B, int ID, int IH, int IW, const int* grid_in, int* nbr,
                       void* stream);
/* out[o] = relu?( (sum_k W[k]^T in[nbr[o][k]]) * scale + shift (+ residual[o]) ); wgt is the
 * spconv layout [K, Cin, Cout] (= [kz,ky,kx,Cin,Cout] flattened).  One launch per layer. */
int al3d_sp_conv_f32(const float* fin, const int* nbr, int K, const float* wgt, int cin, int cout,
                     const float* scale, const float* shift, const float* residual, int relu,
                     float* fout, int n_out, void* stream);
/* Same layer on the fp32 matrix cores (implicit GEMM over 128-row tiles; the rulebook gather
 * is the A-operand address).  wgt_ock is the weight re-packed [Cout, K, Cin].  Channel pairs:
 * 16->32, 32->32, 32->64, 64->64, 64->128, 128->128 (the 16-wide layers stay on the VALU
 * kernel above). */
int al3d_sp_conv_mfma_f32(const float* fin, const int* nbr, int K, const float* wgt_ock, int cin,
                          int cout, const float* scale, const float* shift, const float* residual,
                          int relu, float* fout, int n_out, void* stream);
/* fp32-faithful bf16x6 variant (see al3d_conv2d_nhwc_bf16x6): weights = al3d_split_bf16x3 of
 * the [Cout, K, Cin] packing; same channel pairs as the fp32-MFMA entry point plus 16->16.
 * Three kernel structures, one arithmetic (the MFMA sequence per output element is the same, so
 * their results are bit-identical):
 *   al3d_sp_conv_bf16x6        128-row LDS-staged tile (gathered rows split into LDS like a dense GEMM)
 *   al3d_sp_conv_wave_bf16x6   each wave owns 32 output rows and all output channels; gathered rows
 *                              go straight into its MFMA fragments, only weight slabs pass through LDS
 *   al3d_sp_conv_wave2_bf16x6  the wave kernel with a software-pipelined gather (register ring of
 *                              neighbour indices 2P units ahead and row fragments P units ahead,
 *                              branch-free loads, XCD-contiguous tiles) -- the encoder's default. */
int al3d_sp_conv_bf16x6(const float* fin, const int* nbr, int K, const void* wgt_bf16x3, int cin,
                        int cout, const float* scale, const float* shift, const float* residual,
                        int relu, float* fout, int n_out, void* stream);
int al3d_sp_conv_wave_bf16x6(const float* fin, const int* nbr, int K, const void* wgt_bf16x3, int cin,
                             int cout, const float* scale, const float* shift, const float* residual,
                             int relu, float* fout, int n_out, void* stream);
int al3d_sp_conv_wave2_bf16x6(const float* fin, const int* nbr, int K, const void* wgt_bf16x3, int cin,
                              int cout, const float* scale, const float* shift, const float* residual,
                              int relu, float* fout, int n_out, void* stream);
/* the software-pipelined wave kernel in f16x3 arithmetic (see al3d_conv2d_nhwc_f16x3): weights =
 * al3d_split_f16x3 of [Cout][K][Cin] (two f16 planes), `scale` REQUIRED and carrying 2^-scale_exp;
 * gathered activations must stay below 65504.  Default of the sparse encoder. */
int al3d_sp_conv_wave2_f16x3(const float* fin, const int* nbr, int K, const void* wgt_f16x2, int cin,
                             int cout, const float* scale, const float* shift, const float* residual,
                             int relu, float* fout, int n_out, void* stream);
/* Tiled rulebook for al3d_sp_conv_glds_f16x3: the tables of al3d_sp_subm_table / al3d_sp_down_table with a
 * row pitch of al3d_sp_table_pitch(n) (a multiple of 256; entries of rows >= n are -1), so that the 32
 * indices of one (tap, 32-row tile) form one aligned 128-byte line, plus tile_mask[pitch/32]: bit k set
 * iff tap k has a neighbour for at least one row of the tile (zeroed and filled here).  At most 27 taps.
 * Same rulebook statement as above: geometry.h:25-82,145-194. */
int al3d_sp_table_pitch(int n);
int al3d_sp_subm_table_tiles(const int* coords, int n, int B, int D, int H, int W, const int* grid, int kd,
                             int kh, int kw, int* nbr, int pitch, unsigned* tile_mask, void* stream);
int al3d_sp_down_table_tiles(const int* coords_out, int n_out, const int* ksize, const int* stride,
                             const int* pad, int B, int ID, int IH, int IW, const int* grid_in, int* nbr,
                             int pitch, unsigned* tile_mask, void* stream);
/* The sparse layer (scn.py:331-369 via spconv_ops.h:260-361) with an LDS-DMA row gather: every gathered
 * row is fetched as full 128-byte lines by global_load_lds (8 line lookups per KiB instead of the 64 of
 * fragment-shaped register loads) into a swizzled LDS image, each wave owns several 32-row tiles and
 * walks the (tap, tile) items their tile masks name, a producer wave streams the weight slabs; f16x3
 * arithmetic in al3d_sp_conv_wave2_f16x3's summation order, so results are BIT-IDENTICAL to it.
 * nbr / nbr_pitch / tile_mask: a tiled rulebook (above).  `wgt_image` = al3d_sp_pack_glds_f16x3 of the two
 * f16 planes ([K][Cin/16][2][ceil32(Cout)][16] f16, halves swizzled, LDS image order, zero rows beyond
 * Cout); al3d_sp_pack_glds_f16x3_elems gives its element count (-1: unsupported shape). */
int64_t al3d_sp_pack_glds_f16x3_elems(int cout, int K, int cin);
int al3d_sp_pack_glds_f16x3(const void* planes_f16x2, int cout, int K, int cin, void* out_image, void* stream);
int al3d_sp_conv_glds_f16x3(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                            const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                            const float* residual, int relu, float* fout, int n_out, void* stream);

/* Range-gather form of the same layer (27-tap submanifold only; csrc/spconv_rng.hip): the three kx taps of a
 * (kz, ky) group share ONE staged index range [lo, lo + len) of input rows -- tile_rng [ceil(n_out/32)][9][2] =
 * (lo, len) from al3d_sp_tile_ranges on the same tiled table (len 0: no neighbour; len > 48: the group is gathered
 * row by row).  Same weight image, same contract otherwise; bit-identical to al3d_sp_conv_glds_f16x3 for Cin = 32
 * (with more than one 32-channel chunk the chunks of a group are summed before the next group: last-bit differences,
 * the same error class). */
int al3d_sp_tile_ranges(const int* nbr, int64_t nbr_pitch, int K, int n_out, int* out_rng, void* stream);
int al3d_sp_conv_rng_f16x3(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask,
                           const int* tile_rng, int K, const void* wgt_image, int cin, int cout, const float* scale,
                           const float* shift, const float* residual, int relu, float* fout, int n_out, int io,
                           void* stream);

/* Block-staged form of the same layer (27-tap submanifold, Cin = Cout in {32, 64, 128}; csrc/spconv_blk.hip): a
 * workgroup owns R consecutive output rows and stages the UNION of their 27-tap neighbourhoods once into LDS; the taps
 * read it through 16-bit local indices.  al3d_sp_block_shape gives (R, CAP) of the kernel built for a channel pair;
 * al3d_sp_block_plan builds, from the level's tiled table, per chunk of R rows: hdr [chunks][2] = (staged rows U,
 * flag), rows [chunks][CAP] = the staged row ids (ascending within id windows), loc [chunks][27][R] uint16 = staged
 * position of (tap, row) or 0xffff.  A chunk whose union does not fit (flag 1: no locality in the row order) is
 * gathered tap by tap inside the same kernel.  One plan per table, shared by the level's layers.  Same weight image
 * and contract as al3d_sp_conv_glds_f16x3_io; BIT-IDENTICAL to al3d_sp_conv_wave2_f16x3 in any row order.
 * Replaces spconv's indice_conv for these layers (spconv_ops.h:260-361; call sites det3d/models/backbones/scn.py:349-369). */
int al3d_sp_block_shape(int cin, int cout, int* rows_per_chunk, int* staged_cap);
int al3d_sp_block_plan(const int* nbr, int64_t nbr_pitch, int K, int n_out, int rows_per_chunk, int staged_cap,
                       int* out_hdr, int* out_rows, void* out_loc, void* stream);
int al3d_sp_conv_blk_f16x3(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask,
                           const int* plan_hdr, const int* plan_rows, const void* plan_loc, int K,
                           const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                           const float* residual, int relu, float* fout, int n_out, int io, void* stream);

/* Level-0 layers (16 input channels) on RASTER-ordered rows (csrc/spconv_l0.hip; the layers are
 * det3d/models/backbones/scn.py:331-347, rulebook semantics geometry.h:248-298).  The voxelizer's rows arrive in
 * first-appearance order (the reference's order of example["coordinates"]); inside the encoder the row order of a
 * level is free, so the encoder renumbers level 0 in raster order (b, z, y, x):
 *   al3d_sp_raster_perm: coords [n][4] (b, z, y, x; unique cells; W <= 2048) -> perm [n] (raster position ->
 *     original row) and coords_raster [n][4]; a counting sort over the (b, z, y) lines + a bit-mask rank inside each
 *     line.  workspace >= al3d_sp_raster_perm_workspace_bytes(n, B, D, H).  frame_rows_max > 0 promises frame-sorted
 *     rows (coords[:, 0] ascending) with at most that many rows per frame: up to 65,535 rows, D H <= 43,008 lines and
 *     W <= 1024 the whole sort then runs in LDS, one workgroup per frame; 0: no promise, the general path.
 *     The promise is checked on the device: the first int32 of `workspace` is a STATUS word the caller reads after the
 *     call's stream work (0 = kept; bit 0 = rows not frame-sorted, bit 1 = a frame beyond 65,535 rows).  On a violation
 *     perm / coords_raster still hold a valid permutation (the identity, for the offending rows) -- never an
 *     uninitialised index -- but the rows are not in raster order: treat a non-zero status as an error.
 *   al3d_sp_rows_gather_pad_f32: out[r] = rows[perm[r]] (perm NULL: identity) zero-padded from channels_in to
 *     channels_out (% 8 == 0) channels, as f32 rows or pair rows.
 * In raster order the neighbours of 32 consecutive output rows under the three kx taps of a (kz, ky) group lie in one
 * short contiguous index range, and every live (32-row tile, group) pair of a tiled 27-tap table becomes one ITEM:
 *   al3d_sp_tile_items: first [ntiles + 1] (index of a tile's first item; first[ntiles] = item count), items
 *     [9 ntiles + 1] int4 {lo, len | group << 16 | first-of-tile << 20 | last-of-tile << 21, tile tap mask, tile};
 *     workspace >= al3d_sp_tile_items_workspace_bytes(n_out).  Works on submanifold and strided tables.
 *   al3d_sp_pack_r16_f16x3: f16 planes [2][Cout][27][16] -> the kernel's LDS image [27][2][2 k-halves][Cout][8]
 *     (Cout 16 or 32; al3d_sp_pack_r16_f16x3_elems elements).
 *   al3d_sp_conv_r16_f16x3: the layer as a stream of items per wave: each item's index range is staged once by LDS-DMA
 *     as whole lines and serves three taps; all 27 taps' weights stay in LDS; no barrier in the main loop.  Ranges
 *     longer than the staged window are gathered row by row (any row order is correct, raster order is fast).
 *     f16x3 arithmetic in al3d_sp_conv_wave2_f16x3's summation order: BIT-IDENTICAL to it.  residual only at Cout 16;
 *     io as below; tiles_per_wave <= 0: default. */
int64_t al3d_sp_raster_perm_workspace_bytes(int n, int B, int D, int H);
int al3d_sp_raster_perm(const int* coords, int n, int B, int D, int H, int W, int frame_rows_max, void* workspace,
                        int* perm, int* coords_raster, void* stream);
int al3d_sp_rows_gather_pad_f32(const float* rows, const int* perm, int64_t n, int channels_in, int channels_out,
                                int to_pair, float* out, void* stream);
int64_t al3d_sp_tile_items_workspace_bytes(int n_out);
int al3d_sp_tile_items(const int* nbr, int64_t nbr_pitch, int K, int n_out, const unsigned* tile_mask, void* workspace,
                       int* first, void* items, void* stream);
int64_t al3d_sp_pack_r16_f16x3_elems(int cout);
int al3d_sp_pack_r16_f16x3(const void* planes_f16x2, int cout, void* out_image, void* stream);
int al3d_sp_conv_r16_f16x3(const float* fin, const int* nbr, int nbr_pitch, const void* items, const int* first, int K,
                           const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                           const float* residual, int relu, float* fout, int n_out, int io, int tiles_per_wave,
                           void* stream);

/* Row formats of the sparse encoder's activations (csrc/sp_rows.h).  "pair rows" hold, per 8 channels, the two f16
 * planes the f16x3 arithmetic multiplies with (16 B xh = f16(x), 16 B xl' = f16((x - xh) 2^11)) in the 32 bytes of
 * the 8 floats: a consumer's fragment load is its MFMA operand pair and the split runs once, in the producer's
 * epilogue, instead of once per gathered (row, tap).  io flags: bit 0 = fin is pair rows, bit 1 = write pair rows,
 * bit 2 = residual is pair rows.  A layer fed pair rows gives the bits of the same layer fed the f32 rows they
 * were split from; al3d_sp_rows_convert_f16x3 converts [n][channels] either way (channels % 8 == 0). */
int al3d_sp_rows_convert_f16x3(const float* in, int64_t n, int channels, int to_pair, float* out, void* stream);
int al3d_sp_conv_glds_f16x3_io(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                               const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                               const float* residual, int relu, float* fout, int n_out, int io, void* stream);
int al3d_sp_conv_wave2_f16x3_tiles_io(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask,
                                      int K, const void* wgt_f16x2, int cin, int cout, const float* scale,
                                      const float* shift, const float* residual, int relu, float* fout, int n_out,
                                      int io, void* stream);
/* al3d_sp_conv_wave2_f16x3 on a tiled rulebook: the offsets a wave needs come from tile_mask[tile] (one word) instead
 * of a scan of the tile's 27 x 32 table entries; same arithmetic, same bits. */
int al3d_sp_conv_wave2_f16x3_tiles(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                                   const void* wgt_f16x2, int cin, int cout, const float* scale, const float* shift,
                                   const float* residual, int relu, float* fout, int n_out, void* stream);
/* planes [3][count] bf16 -> f32 [count], exact (inverse of al3d_split_bf16x3) */
int al3d_merge_bf16x3(const void* planes_bf16x3, int64_t count, float* out, void* stream);
/* dense(): out NHWC [B,H,W,C*D] with channel = c*D + z (== .dense().view(N, C*D, H, W));
 * out must be zero-filled. */
int al3d_sp_to_dense_nhwc(const float* feat, const int* coords, int n, int C, int B, int D, int H,
                          int W, float* out, void* stream);

/* ---------------------------------------------------------------- detector: dense */

/* NHWC f32 convolution on the fp32 matrix cores with fused per-channel
 * y = conv*scale[c] + shift[c] (+ReLU).  in [B,H,W,Cin]; wgt [Cout,k*k,Cin];
 * out [B,OH,OW,ldc] written at channels [coff, coff+Cout).  Cin % 32 == 0.
 * Replaces the Conv2d+BatchNorm2d(eval)+ReLU triples of RPN
 * (det3d/models/necks/rpn.py:124-142) and the 1x1 task heads
 * (det3d/models/bbox_heads/mg_head.py:215-231; scale=NULL, shift=bias, relu=0). */
int al3d_conv2d_nhwc_f32(const float* in, const float* wgt, const float* scale, const float* shift,
                         float* out, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                         int pad, int ldc, int coff, int relu, void* stream);

/* ConvTranspose2d(kernel 2, stride 2) + scale/shift (+ReLU), NHWC f32.
 * in [B,H,W,Cin]; wgt [Cout,4,Cin] with tap = dy*2+dx; out [B,2H,2W,ldc].
 * Replaces RPN deblock 1 (det3d/models/necks/rpn.py:79-93). */
int al3d_deconv2x2_nhwc_f32(const float* in, const float* wgt, const float* scale,
                            const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                            int ldc, int coff, int relu, void* stream);

/* fp32-faithful variants on the bf16 matrix cores ("bf16x6"): every fp32 operand is split
 * exactly into three bf16 pieces and each product is formed from the six partial products of
 * order <= 2 (dropped terms < 2^-25 relative), fp32 accumulation.  Same contract as the two
 * entry points above except that the weights are pre-split with al3d_split_bf16x3()
 * (f32 [count] -> bf16 [3][count], planes hi/mid/lo) and Cin % 16 == 0. */
int al3d_split_bf16x3(const float* w, int64_t count, void* out_bf16x3, void* stream);
int al3d_conv2d_nhwc_bf16x6(const float* in, const void* wgt_bf16x3, const float* scale,
                            const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                            int ksize, int stride, int pad, int ldc, int coff, int relu, void* stream);
int al3d_deconv2x2_nhwc_bf16x6(const float* in, const void* wgt_bf16x3, const float* scale,
                               const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                               int ldc, int coff, int relu, void* stream);

/* fp32-class variants on the f16 matrix cores ("f16x3", the default of the dense neck + head):
 * activations are split while staged into xh = f16(x), xl = f16((x - xh) * 2^11); weights are
 * pre-split once with al3d_split_f16x3() -- f32 [count], multiplied by 2^scale_exp chosen by the
 * caller so that max|w * 2^scale_exp| <= 2^14 -- into f16 [2][count] planes
 * (wh, wl = ws - wh); x*ws = xh*wh + xh*wl + xl*(wh * 2^-11) in ONE fp32 accumulator, dropped term
 * <= 2^-24 relative.  `scale` is REQUIRED and must already carry the factor
 * 2^-scale_exp (exact).  Valid for |activation| < 65504 (larger values give inf/NaN outputs --
 * use the bf16x6 entries for the full fp32 range).  Otherwise the contract of
 * al3d_conv2d_nhwc_f32 / al3d_deconv2x2_nhwc_f32 (same reference lines), Cin % 16 == 0. */
int al3d_split_f16x3(const float* w, int64_t count, int scale_exp, void* out_f16x2, void* stream);
int al3d_conv2d_nhwc_f16x3(const float* in, const void* wgt_f16x2, const float* scale,
                           const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                           int ksize, int stride, int pad, int ldc, int coff, int relu, void* stream);
int al3d_deconv2x2_nhwc_f16x3(const float* in, const void* wgt_f16x2, const float* scale,
                              const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                              int ldc, int coff, int relu, void* stream);
/* 3x3 / stride 1 / pad 1 layers (11 of the neck's 15 launches, rpn.py:66-113): same arithmetic,
 * weights re-arranged once by al3d_pack_f16x3_frag() from the [2][Cout][9][Cin] planes into MFMA
 * fragment order [2][Cout/32][Cin/16][9][64 lanes][8] so that every wave streams its B operands
 * straight from L2 into registers (no LDS staging of weights).  Cout % 128 == 0, Cin % 32 == 0. */
int al3d_pack_f16x3_frag(const void* planes_f16x2, int Cout, int Cin, void* out_frag, void* stream);
/* The same layer on v_mfma_f32_16x16x32_f16 (same flops at less power: the 3x3 kernel is power-limited): weights in
 * 16x16x32 fragment order [2][Cout/16][Cin/32][9][64][8] from al3d_pack_f16x3_frag16 (Cout % 128 == 0, Cin % 64 == 0).
 * Same three products per MAC into one fp32 accumulator, 32 input channels per instruction instead of 16: fp32-class
 * like the others, not bit-identical to them. */
int al3d_pack_f16x3_frag16(const void* planes_f16x2, int Cout, int Cin, void* out_frag, void* stream);
int al3d_conv3x3_nhwc_f16x3_frag16(const float* in, const void* wgt_frag16, const float* scale, const float* shift,
                                   float* out, int B, int H, int W, int Cin, int Cout, int ldc, int coff, int relu,
                                   void* stream);
int al3d_conv3x3_nhwc_f16x3_frag(const float* in, const void* wgt_frag, const float* scale,
                                 const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                                 int ldc, int coff, int relu, void* stream);
/* ... writing pair pixels (io = 2; csrc/sp_rows.h, see al3d_conv2d_nhwc_f16x3_dma) for a consumer on the LDS-DMA kernel;
 * io = 0: exactly the call above. */
int al3d_conv3x3_nhwc_f16x3_frag_io(const float* in, const void* wgt_frag, const float* scale,
                                    const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                                    int ldc, int coff, int relu, int io, void* stream);
/* Every other geometry (stride-2 block entry, 1x1 deblock, 2x2 deconvolution, fused 1x1 head:
 * rpn.py:66-113, mg_head.py:215-231) with the weights streamed the same way: activations staged
 * through LDS per (tap, 16-channel chunk), B fragments from al3d_pack_f16x3_bstream(): planes
 * [2][Cout][taps][Cin] -> [2][ceil(Cout/128)*4][taps][Cin/16][64][8] (zero rows beyond Cout;
 * al3d_pack_f16x3_bstream_elems() f16 elements).  Cin % 16 == 0; deconv: taps = 4, tap = dy*2+dx.
 * Bit-identical to al3d_conv2d_nhwc_f16x3 / al3d_deconv2x2_nhwc_f16x3 on the plane layout. */
int64_t al3d_pack_f16x3_bstream_elems(int Cout, int taps, int Cin);
int al3d_pack_f16x3_bstream(const void* planes_f16x2, int Cout, int taps, int Cin, void* out_frag,
                            void* stream);
int al3d_conv2d_nhwc_f16x3_bstream(const float* in, const void* wgt_frag, const float* scale,
                                   const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                                   int ksize, int stride, int pad, int ldc, int coff, int relu, void* stream);
int al3d_deconv2x2_nhwc_f16x3_bstream(const float* in, const void* wgt_frag, const float* scale,
                                      const float* shift, float* out, int B, int H, int W, int Cin, int Cout,
                                      int ldc, int coff, int relu, void* stream);

/* The same geometries with BOTH operands fetched by LDS-DMA (global_load_lds_dwordx4) into a ring of stages that
 * runs 3 steps ahead: raw fp32 activation tile + weight tile from al3d_pack_f16x3_dma(): planes
 * [2][Cout][taps][Cin] -> [ceil(Cout/128)][taps][Cin/16][2][128][16] f16 with the 16-byte halves of a row swizzled
 * (the LDS image of a step; al3d_pack_f16x3_bstream_elems() elements).  The activation split runs on the fragment.
 * gap_part may be NULL; otherwise as al3d_conv2d_nhwc_f16x3_gap.  Bit-identical to al3d_conv2d_nhwc_f16x3 /
 * al3d_deconv2x2_nhwc_f16x3 (and their _gap partials).  io: pixel formats, as the sparse rows (csrc/sp_rows.h;
 * al3d_sp_rows_convert_f16x3 converts [B*H*W][C] either way): bit 0 = `in` holds pair pixels (per 8 channels xh[8] |
 * xl'[8]: the fragment is the operand pair, no split in the kernel), bit 1 = write pair pixels (needs Cout, ldc, coff
 * multiples of 8); the GAP partials are always sums of the f32 values. */
int al3d_pack_f16x3_dma(const void* planes_f16x2, int Cout, int taps, int Cin, void* out_image, void* stream);
int al3d_conv2d_nhwc_f16x3_dma(const float* in, const void* wgt_image, const float* scale, const float* shift,
                               float* out, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad,
                               int ldc, int coff, int relu, float* gap_part, int gap_parts, int io, void* stream);
int al3d_deconv2x2_nhwc_f16x3_dma(const float* in, const void* wgt_image, const float* scale, const float* shift,
                                  float* out, int B, int H, int W, int Cin, int Cout, int ldc, int coff, int relu,
                                  float* gap_part, int gap_parts, int io, void* stream);

/* Fused global average pooling (feature_selector.py:68-71 tap): the two deblock launches of
 * the neck (1x1 conv and 2x2 transposed conv, rpn.py:124-142) can also emit, per workgroup and wave row, the channel
 * sums of the values they store -- gap_part [B][gap_parts][ldc], gap_parts >= al3d_gap_parts_count(OH, OW, deconv), the
 * launch fills its own count of slots and leaves the rest untouched (zero them when launches share a buffer) -- so the
 * [B,128,128,512] map is not read again (33.5 MB per frame); al3d_gap_reduce_parts_f32 adds the parts in ascending order
 * and divides by count = OH*OW.  Deterministic; differs from al3d_gap_nhwc_f32's W-then-H order in the last bits. */
int al3d_gap_parts_count(int OH, int OW, int deconv);
int al3d_conv2d_nhwc_f16x3_gap(const float* in, const void* wgt_f16x3, const float* scale, const float* shift,
                               float* out, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad,
                               int ldc, int coff, int relu, float* gap_part, int gap_parts, void* stream);
int al3d_deconv2x2_nhwc_f16x3_gap(const float* in, const void* wgt_f16x3, const float* scale, const float* shift,
                                  float* out, int B, int H, int W, int Cin, int Cout, int ldc, int coff, int relu,
                                  float* gap_part, int gap_parts, void* stream);
int al3d_gap_reduce_parts_f32(const float* gap_part, int B, int parts, int C, int64_t count, float* out, void* stream);
/* BEV embedding: mean over W then over H of an NHWC map, [B,H,W,C] -> [B,C].
 * Replaces `fpn_feats[-1].mean(-1).mean(-1)` (det3d/selectors/feature_selector.py:68-71). */
int64_t al3d_gap_workspace_bytes(int B, int H, int C);
int al3d_gap_nhwc_f32(const float* x, int B, int H, int W, int C, float* out, void* workspace,
                      void* stream);

/* ---------------------------------------------------------------- camera branch: BEV pooling (f4)
 * bevfusion/mmdet3d/models/vtransforms/base.py:127-163 (`bev_pool`: cell = ((geom - (bx - dx/2)) / dx).long(),
 * points outside the grid dropped) + ops/bev_pool/bev_pool.py:82-97 + src/bev_pool_cuda.cu:21-44 (sum of the points
 * of a cell).  geom [P,3] f32 lidar-frame positions (points of sample b: [b*P/B, (b+1)*P/B)); lo = bx - dx/2 and dx
 * as float32[3], nx int[3]; out [B, nx0, nx1, nx2*C] f32 channels-last with channel = iz*C + c (the reference's
 * [B,C,D,H,W] -> cat(unbind(2),1)), every cell written (empty cells 0).  Summation order: ascending point index
 * (deterministic; the reference's argsort order is implementation-defined).
 * _lss: the Lift-Splat outer product fused in (depth_lss.py:92-97): x[p] = depth[p] * ctx[pixel(p)] with
 * depth [BN,D,fH,fW], ctx [BN,fH,fW,C] channels-last, p = ((bn*D + d)*fH + h)*fW + w; the [BN,D,fH,fW,C] tensor
 * is never materialised. */
/* Frustum geometry of the Lift-Splat transform (vtransforms/base.py:79-122) for BN cameras: frustum [D*fH*fW,3]
 * (u, v, depth), cam_rows [BN,44] f32 on device = inv(post_rot)[9] | post_trans[3] | camera2lidar_rot inv(intrins)[9]
 * | camera2lidar_trans[3] | extra_rot[9] | extra_trans[3] | has_extra_rot | has_extra_trans | pad[6];
 * geom [BN*D*fH*fW,3] lidar-frame positions. */
int64_t al3d_lss_geometry_workspace_bytes(int BN);
int al3d_lss_geometry_f32(const float* frustum, int64_t points_per_camera, const float* cam_rows, int BN, float* geom,
                          void* workspace, void* stream);

/* Lidar depth image of the depth-aware LSS transform (vtransforms/base.py:225-262, BaseDepthTransform.forward): the
 * points of one sample, taken back through the lidar augmentation, projected into every camera and through the image
 * augmentation, write their depth to depth [ncam][iH][iW] at the truncated pixel; of several points on one pixel the
 * LAST in point order stays (the reference's indexed assignment on the CPU).  cam_rows [ncam][24] = lidar2image[:3,:3]
 * | lidar2image[:3,3] | img_aug[:3,:3] | img_aug[:3,3]; aug_rows [12] = inverse(lidar_aug[:3,:3]) | lidar_aug[:3,3]
 * (device memory).  workspace: al3d_lss_depth_image_workspace_bytes. */
int64_t al3d_lss_depth_image_workspace_bytes(int ncam, int iH, int iW);
int al3d_lss_depth_image_f32(const float* points, int64_t npts, int stride, const float* cam_rows, int ncam,
                             const float* aug_rows, int iH, int iW, float* depth, void* workspace, void* stream);
/* depth_lss.py:93-96: softmax over the D depth logits of the depth net's channels-last output y [BN][fH][fW][ldy] (logits
 * in channels 0 .. D-1), written as [BN][D][fH][fW] probabilities for the pooling; D <= 256. */
int al3d_lss_depth_softmax_f32(const float* y, int BN, int fH, int fW, int D, int ldy, float* out, void* stream);
/* necks/generalized_lss.py:88-101, one top-down step of the LSS-FPN: out [N][H][W][C1 + C2] (channels-last) =
 * cat(lat [N][H][W][C1], bilinear upsample (align_corners = True, torch's weights) of src [N][h][w][C2]); C1, C2 % 4 == 0. */
int al3d_lss_upsample_cat_f32(const float* lat, const float* src, int N, int H, int W, int C1, int h, int w, int C2, float* out,
                              void* stream);
/* ... with upsample_cfg's align_corners as a parameter (the class default is True; the shipped swint configs set false:
 * bevfusion/configs/nuscenes/det/transfusion/secfpn/camera+lidar/default.yaml:16-18): 0 = torch's half-pixel source
 * positions max(0, (o + 0.5) in / out - 0.5). */
int al3d_lss_upsample_cat_mode_f32(const float* lat, const float* src, int N, int H, int W, int C1, int h, int w, int C2,
                                   int align_corners, float* out, void* stream);
/* Channel concatenation of two channels-last maps, out [N][H][W][Ca + Cb] = cat(a, b) (vtransforms/depth_lss.py:84 before the
 * depth net, fusers/conv.py:24 before the fuser); a_hw_swapped: a is stored [N][W][H][Ca] (the view transform's [x, y] map)
 * and is transposed on the way.  Ca, Cb % 4 == 0. */
int al3d_cat2_nhwc_f32(const float* a, const float* b, int N, int H, int W, int Ca, int Cb, int a_hw_swapped, float* out,
                       void* stream);
/* depth_lss.py:38-44, the first two layers of `dtransform` as one kernel: Conv2d(1, 8, 1) + BN + ReLU -> Conv2d(8, 32, 5,
 * stride 4, padding 2) + BN + ReLU on the depth image [BN][iH][iW] -> out [BN][oH][oW][32] (channels-last, oH =
 * (iH - 1) / 4 + 1).  p0 = [w0[8] | scale0[8] | shift0[8]] with layer 0 = relu((w0 d) scale0 + shift0) (bias and BN folded:
 * shift0 = bn_shift + bias * bn_scale); w1 [25][8][32] = conv.weight[co][c][ky][kx] at [(ky*5+kx)][c][co];
 * p1 = [scale1[32] | shift1[32]].  fp32 FMA chain over (tap, channel); zero padding applies to layer 0's OUTPUT. */
int al3d_lss_dtransform01_f32(const float* depth, int BN, int iH, int iW, const float* p0, const float* w1, const float* p1,
                              float* out, void* stream);
int64_t al3d_bev_pool_workspace_bytes(int64_t n_points, int64_t n_cells);
int al3d_bev_pool_f32(const float* x, const float* geom, int64_t n_points, int C, int B, const float* lo,
                      const float* dx, const int* nx, float* out, void* workspace, void* stream);
int al3d_bev_pool_lss_f32(const float* depth, const float* ctx, const float* geom, int BN, int D, int fH, int fW,
                          int C, int B, const float* lo, const float* dx, const int* nx, float* out,
                          void* workspace, void* stream);
/* The same pooling in two halves: the plan (every frustum point's cell, every cell's members in ascending point order)
 * depends on the geometry only, i.e. on the calibration matrices; a sweep over a fixed camera rig builds it once
 * (al3d_bev_pool_plan, into a workspace of al3d_bev_pool_workspace_bytes) and applies it to every batch's (depth,
 * context) maps (al3d_bev_pool_lss_apply_f32).  plan + apply == al3d_bev_pool_lss_f32 bit for bit. */
int al3d_bev_pool_plan(const float* geom, int64_t n_points, int B, const float* lo, const float* dx, const int* nx,
                       void* workspace, void* stream);
int al3d_bev_pool_lss_apply_f32(const float* depth, const float* ctx, int BN, int D, int fH, int fW, int C, int B,
                                const int* nx, const void* plan_workspace, float* out, void* stream);


/* ---------------------------------------------------------------- streaming file loader (a1 / f2)
 * Host reader pool (csrc/reader.cpp, pthreads): replaces the reference's DataLoader worker processes
 * (det3d/datasets/loader/build_loader.py:23-59) for the part that touches files -- read_file / read_sweep's
 * np.fromfile (det3d/datasets/pipelines/loading.py:17-24,27-36).  plan() stats the files (rows = whole
 * 20-byte x,y,z,intensity,ring records); submit() starts reading file i's rows to dst + 20*row_off[i]
 * (dst is the caller's pinned staging buffer) on the pool and returns a job id; wait() blocks until that
 * job's files have landed (an I/O error of a worker is reported here).  Thread-safe per reader. */
typedef struct al3d_reader al3d_reader;
int al3d_reader_create(int n_threads, al3d_reader** out);
void al3d_reader_destroy(al3d_reader* reader);
int64_t al3d_reader_plan(const char* const* paths, int n_files, int64_t* rows_out);
int al3d_reader_submit(al3d_reader* reader, const char* const* paths, int n_files, const int64_t* row_off,
                       const int64_t* rows, void* dst, int64_t dst_bytes);
int al3d_reader_wait(al3d_reader* reader, int job_id);
/* al3d_merge_sweeps_f32 for the files of n_frames frames back to back (loading.py:98-126 per frame):
 * is_key[f] marks a frame's key file (kept whole, time 0), frame_first_file [n_frames+1] the file index each
 * frame starts at; out = the frames' clouds back to back, out_frame_off [n_frames+1] int64 their first
 * points.  Same arithmetic as al3d_merge_sweeps_f32 (bit-identical per frame); workspace as
 * al3d_merge_sweeps_workspace_bytes(total_rows). */
int al3d_merge_sweeps_batch_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                const double* xform, const unsigned char* has_xform, const double* time_lag,
                                const unsigned char* is_key, const int* frame_first_file, int n_frames,
                                float min_distance, float* out, int64_t* out_frame_off, void* workspace,
                                void* stream);
/* The same with the transform rule as a parameter: 0 = det3d's loader (above); 1 = BEVFusion's LoadPointsFromMultiSweeps
 * (bevfusion/mmdet3d/datasets/pipelines/loading.py:84-237): `p[:, :3] = p[:, :3] @ R.T` rounds the float64 rotation to
 * float32, `p[:, :3] += t` adds the float64 translation to the rounded value and rounds again (xform rows = [R | t]);
 * time column = float32(ts - sweep_ts) (time_lag); key frame unfiltered with time 0; the key-frame copies the reference
 * appends for an EMPTY sweep list (pad_empty_sweeps) are ordinary files with has_xform 0 and time lag 0. */
int al3d_merge_sweeps_batch_rule_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                     const double* xform, const unsigned char* has_xform, const double* time_lag,
                                     const unsigned char* is_key, const int* frame_first_file, int n_frames,
                                     float min_distance, int rule, float* out, int64_t* out_frame_off, void* workspace,
                                     void* stream);
/* ... followed by the test pipeline's PointsRangeFilter (bevfusion/mmdet3d/datasets/pipelines/transforms_3d.py:503-525 on
 * core/points/base_points.py:208-232; bevfusion/configs/nuscenes/default.yaml:233-235): point_range = HOST float[6]
 * (x_min, y_min, z_min, x_max, y_max, z_max) or NULL; a merged (transformed, float32) point stays when it lies STRICTLY
 * inside the range on x, y and z. */
int al3d_merge_sweeps_batch_range_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                      const double* xform, const unsigned char* has_xform, const double* time_lag,
                                      const unsigned char* is_key, const int* frame_first_file, int n_frames,
                                      float min_distance, int rule, const float* point_range, float* out,
                                      int64_t* out_frame_off, void* workspace, void* stream);

/* Split JPEG decoding for configs[4] from files (csrc/jpeg_host.cpp + csrc/jpeg.hip).  The reference decodes every camera
 * frame with Pillow's Image.open in DataLoader workers (bevfusion/mmdet3d/datasets/pipelines/loading.py:19-58), i.e. with
 * libjpeg-turbo at its defaults; here the Huffman entropy decoding stays on host threads and everything after it runs on the
 * device, bit-identical to Pillow's RGB bytes (accurate integer IDCT, fancy chroma upsampling, 16-bit fixed-point YCbCr -> RGB).
 *   al3d_jpeg_header: HOST.  Parse the markers of a baseline JPEG in memory -> info [AL3D_JPEG_INFO_INTS] = {width, height,
 *     ncomp, h[3], v[3], mcus_x, mcus_y, blocks_w[3], blocks_h[3], block_offset[3], total_blocks, restart_interval, max_h,
 *     max_v, 0...} and quant [3][64] uint16 (natural order, per component).  AL3D_EINVAL (with the reason in
 *     al3d_last_error) for anything but 8-bit baseline Huffman, one interleaved scan, 1 or 3 components, sampling factors 1 / 2
 *     with the luma at the maximum: the caller then uses another decoder for that file.
 *   al3d_jpeg_entropy_decode: HOST, thread-safe.  -> coefs [total_blocks][64] int16: quantised coefficients in natural order,
 *     component after component, each a plane of blocks_h x blocks_w blocks (whole MCUs).
 *   al3d_jpeg_idct_rgb_u8: DEVICE.  nimg images of ONE geometry (info; coefs [nimg][total_blocks][64], quant [nimg][3][64])
 *     -> out_rgb [nimg][height][width][3] uint8; workspace >= al3d_jpeg_workspace_bytes(info, nimg) (the component planes). */
#define AL3D_JPEG_INFO_INTS 32
int al3d_jpeg_header(const unsigned char* data, int64_t nbytes, int* info, unsigned short* quant);
int al3d_jpeg_entropy_decode(const unsigned char* data, int64_t nbytes, short* coefs, int64_t coef_blocks);
int64_t al3d_jpeg_workspace_bytes(const int* info, int nimg);
int al3d_jpeg_idct_rgb_u8(const short* coefs, const unsigned short* quant, const int* info, int nimg,
                          unsigned char* out_rgb, void* workspace, void* stream);

/* TransFusion query initialisation (csrc/proposals.hip; bevfusion/mmdet3d/models/heads/bbox/transfusion.py:236-275):
 * heat_logits [B][H][W][C] channels-last (the heat-map head's output) -> the P best (class, cell) pairs among the k x k
 * local maxima of sigmoid(heat) (interior cells only; classes of free_class_mask keep every cell), ties broken by the
 * smaller flat index c * HW + cell (the reference's argsort leaves them unspecified), in descending score order:
 * top_class / top_cell [B][P] int64, query_heatmap_score [B][C][P] (the masked scores of each winning cell),
 * query_feat [B*P][hidden] = tokens[b * HW + cell] + class_cols[class] + class_bias (class_cols [C][hidden] = the
 * class-encoding Conv1d's weight columns), query_pos [B*P][2] = bev_pos[cell].  P <= 256, C <= 32. */
int64_t al3d_tf_proposals_workspace_bytes(int B, int H, int W, int C);
int al3d_tf_proposals_f32(const float* heat_logits, int B, int H, int W, int C, int nms_kernel, unsigned free_class_mask,
                          int P, const float* tokens, int hidden, const float* bev_pos, const float* class_cols,
                          const float* class_bias, void* workspace, int64_t* top_class, int64_t* top_cell,
                          float* query_heatmap_score, float* query_feat, float* query_pos, void* stream);

/* Camera images of a BEVFusion sample (csrc/images.hip): the test branch of the reference's image pipeline --
 * LoadMultiViewImageFromFiles (loading.py:19-83; decoding stays on the host), ImageAug3D (transforms_3d.py:26-122:
 * img.resize(resize_dims) = PIL's BICUBIC convolution resize on 8-bit pixels, Pillow pinned at 8.4.0 by
 * bevfusion/README.md:71; img.crop(crop); no flip, rotate(0)), ImageNormalize (transforms_3d.py:903-920: ToTensor +
 * Normalize in float32).
 *   al3d_image_resample_ksize / _coeffs (HOST functions): PIL's per-coordinate filter windows, bounds [out][2] = (first
 *     source index, taps) and 22-bit fixed-point weights [out][ksize]; filter 2 = bilinear, 3 = bicubic.
 *   al3d_image_aug_normalize_u8: imgs [nimg][H][W][3] u8 RGB (device) -> out [nimg][fH][fW][3] f32 channels-last =
 *     ((resize to (rH, rW))[crop_y : crop_y + fH, crop_x : crop_x + fW] / 255 - mean) / std, and / or out_u8 = the 8-bit
 *     crop (either may be NULL).  mean3 / std3 are HOST pointers.  row_first / row_count: the source rows the crop's
 *     vertical windows cover (v_bounds[crop_y] .. v_bounds[crop_y + fH - 1] + taps); workspace >=
 *     al3d_image_aug_workspace_bytes(nimg, row_count, fW).  Bit-identical to PIL's resize + crop and to the torch float32
 *     expression. */
int al3d_image_resample_ksize(int in_size, int out_size, int filter);
int al3d_image_resample_coeffs(int in_size, int out_size, int filter, int* bounds, int* coeffs);
int64_t al3d_image_aug_workspace_bytes(int nimg, int rows, int out_w);
int al3d_image_aug_normalize_u8(const unsigned char* imgs, int nimg, int H, int W, int rH, int rW, int crop_x, int crop_y,
                                int fH, int fW, const int* h_bounds, const int* h_coeffs, int h_ksize, const int* v_bounds,
                                const int* v_coeffs, int v_ksize, const float* mean3, const float* std3, int row_first,
                                int row_count, void* workspace, float* out, unsigned char* out_u8, void* stream);


/* 3x3 / stride 1 / pad 1 as Winograd F(2x2, 3x3) in f16x3 arithmetic (det3d/models/necks/rpn.py:66-113: the eleven
 * stride-1 3x3 layers of the SECOND neck): 2.25 x fewer matrix-core products than the direct kernels above, the input
 * and output transforms in fp32.  wgt_wino = al3d_pack_f16x3_wino() of the al3d_split_f16x3 planes [2][Cout][16][Cin] of
 * U = G g G^T (position a*4 + c; computed in float64 by the host); Cin % 16 == 0, Cout % 64 == 0; io = 0 or 2 (pair
 * pixels out).  fp32-class like the direct kernels, not bit-identical to them. */
int al3d_pack_f16x3_wino(const void* planes_f16x2, int Cout, int Cin, void* out_frag, void* stream);
int al3d_conv3x3_nhwc_f16x3_wino(const float* in, const void* wgt_wino, const float* scale, const float* shift, float* out,
                                 int B, int H, int W, int Cin, int Cout, int ldc, int coff, int relu, int io, void* stream);

/* ---------------------------------------------------------------- token matrices (Swin-T image backbone, BASELINE configs[4])
 * The reference configures mmdet 2.20.0's SwinTransformer (bevfusion/configs/nuscenes/det/transfusion/secfpn/
 * camera+lidar/swint_v0p075/default.yaml:  embed_dims 96, depths [2,2,6,2], num_heads [3,6,12,24], window_size 7),
 * which is not in the reference tree: these entry points implement the published block
 *     x += proj(W-MSA(LN(x)));  x += fc2(GELU(fc1(LN(x))))      (Liu et al., ICCV 2021)
 * and replace, per call, what torch would run as nn.LayerNorm / nn.Linear / softmax / matmul / roll / window
 * partition.  "pair rows" = the f16x3 operand format of csrc/sp_rows.h (same bytes per row as f32).
 *
 * al3d_tok_layernorm_f32: out[i] = LN(concat_{g<G} x[rowmap[i*G+g]]) * gamma + beta over G*C channels (biased variance,
 *   eps inside the root).  rowmap null = identity; entry -1 = absent piece: zeros BEFORE the statistics (patch merging's
 *   map padding), or with zero_out (G == 1) a zero OUTPUT row (the window padding follows norm1).  C % 8 == 0,
 *   G in {1, 4}, G*C <= 1536.
 * al3d_tok_linear_f16x3: out[rowmap[m]] = act((a[m] . W^T) * scale + bias) + residual[rowmap[m]] for m < M; W as
 *   al3d_split_f16x3 planes [2][N][1][K] packed by al3d_pack_f16x3_dma; scale = 2^-s of the split (times a folded BatchNorm scale, if any); act 0 none / 1 exact
 *   (erf) GELU / 2 ReLU; rowmap null = identity, -1 = row dropped; residual (f32 rows, pitch ldr) may alias out.  K % 16 == 0,
 *   N % 4 == 0 (8 for pair output).
 * al3d_tok_window_attention_f32: qkv [nwin*49][3C] (q | k | v, each [heads][32]) -> softmax(q scale k^T + B + mask) v,
 *   [nwin*49][C]; B = table[(yq-yk+6)*13 + (xq-xk+6)][head]; mask = -100 between tokens of different shifted-window
 *   regions, derived from the window's position in its win_rows x win_cols grid and `shift` (0 = none).
 * al3d_tok_window_attention_tokens_f32: the same attention with qkv / out in TOKEN order ([B*H*W][3C] -> [B*H*W][C]): the
 *   cyclic shift, the padding to multiples of 7 and the window partition are evaluated from the window's position, a padded
 *   position's q / k / v row is bias_qkv [3C] (what the qkv layer makes of the zero row the padding puts after the norm) and
 *   its output is cropped -- so LN1, the qkv and the projection GEMMs run on the H x W tokens, not on the padded windows
 *   (16 x 44 tokens pad to 21 x 49: 1.46x the rows; 8 x 22 to 14 x 28: 2.2x).
 * al3d_tok_mlp_f16x3: the MLP half of a block as one kernel, x[t] += fc2(GELU(fc1(LN(x[t])))) in place, for C = 96 / 192
 *   (stages 0-1, whose separate LN / fc1 / fc2 launches are bandwidth-bound; the [T, hidden] activation stays in registers).
 *   image (al3d_tok_mlp_image_bytes): C = 96 (32 x 32 x 16 products, 32 tokens per wave): per 32 hidden units t the MFMA
 *   A-operand fragments [64 lanes][8 halves] of
 *   fc1.weight[32t + lane%32][16kc + 8(lane/32) + e], kc < C/16, planes (wh, wl) of al3d_split_f16x3, then of
 *   fc2.weight[32u + lane%32][32t + 16q + 8(e/4) + 4(lane/32) + e%4], u < C/32, q < 2, planes (wh, wl);
 *   C = 192 (16 x 16 x 32 products, 16 tokens per wave): per 32 hidden units t the fragments of
 *   fc1.weight[32t + 16j + lane%16][32ks + 8(lane/16) + e], j < 2, ks < C/32, planes (wh, wl), then of
 *   fc2.weight[16u + lane%16][32t + (e < 4 ? 4(lane/16) + e : 16 + 4(lane/16) + e - 4)], u < C/16, planes (wh, wl);
 *   scale1 / scale2 = 2^-s of the two splits; bias1 [hidden], bias2 [C]; hidden % 32 == 0.
 * al3d_tok_attn_block_f16x3: the attention half of a block as one kernel, x += proj(W-MSA(LN(x))) in place on the B maps of
 *   H x W token rows x [B*H*W][C], for C = 96 / 192 (stages 0-1: their LN1 / qkv / attention / proj launches are
 *   bandwidth-bound; q, k, v and the attention output never exist in memory).  The cyclic shift by `shift`, the padding to
 *   multiples of 7 (after the norm: a padded position is a zero row, a key like any other, never written back) and the
 *   window partition are evaluated from the window's position; two waves per (window, 32-channel head).
 *   image (al3d_tok_attn_block_image_bytes): per head h the MFMA fragments [C/16][2 planes][64 lanes][8] of qkv.weight
 *   rows C + 32h + lane%32 (k), 2C + 32h + lane%32 (v), 32h + lane%32 (q) and of proj.weight rows 32h + lane%32,
 *   element = column 16kc + 8(lane/32) + e, planes (wh, wl) of al3d_split_f16x3 (one split per matrix: scale_qkv,
 *   scale_proj = 2^-s); bias_qkv [3C], bias_proj [C]; table / shift / attn_scale as al3d_tok_window_attention_f32. */
/* img [B][H][W][3] f32 -> patch rows [B*ceil(H/4)*ceil(W/4)][48], k = (ky*4 + kx)*3 + c, zeros beyond the image: the A
 * matrix of the 4x4 / stride-4 patch embedding (mmdet PatchEmbed: Conv2d(3, 96, 4, 4)) as a token GEMM with K = 48. */
int al3d_tok_patch_rows_f32(const float* img, int B, int H, int W, int out_pair, float* out, void* stream);
/* The whole patch embedding as one kernel for embed dim 96 (mmdet PatchEmbed: Conv2d(3, 96, 4, 4) + LayerNorm(96)):
 * out[b*TH*TW + ty*TW + tx][:] = LN(projection(patch)) (gamma null: no LayerNorm), img channels-last, W % 4 == 0 (rows of a
 * patch are read as aligned 16-byte pieces; pixels beyond H are zero).  image (al3d_tok_patch_embed_image_bytes): the MFMA
 * A fragments [3 tiles][3 steps][2 planes][64 lanes][8] of projection.weight reordered to k = (ky*4 + kx)*3 + c:
 * element = w[32u + lane%32][16s + 8(lane/32) + e], planes (wh, wl) of al3d_split_f16x3, scale = 2^-s. */
int64_t al3d_tok_patch_embed_image_bytes(void);
int al3d_tok_patch_embed_f16x3(const float* img, int B, int H, int W, const void* image, float scale, const float* bias,
                               const float* gamma, const float* beta, float eps, float* out, void* stream);
int al3d_tok_layernorm_f32(const float* x, const int* rowmap, int64_t rows_out, int C, int G, int zero_out,
                           const float* gamma, const float* beta, float eps, int out_pair, float* out, void* stream);
int64_t al3d_tok_mlp_image_bytes(int C, int hidden);
int al3d_tok_mlp_f16x3(float* x, int64_t T, int C, int hidden, const float* gamma, const float* beta, float eps,
                       const void* image, float scale1, const float* bias1, float scale2, const float* bias2, void* stream);
int64_t al3d_tok_attn_block_image_bytes(int C);
int al3d_tok_attn_block_f16x3(float* x, int B, int H, int W, int C, int shift, const float* gamma, const float* beta,
                              float eps, const void* image, float scale_qkv, const float* bias_qkv, float scale_proj,
                              const float* bias_proj, const float* table, float attn_scale, void* stream);
int al3d_tok_linear_f16x3(const float* a, int a_pair, const void* wgt_image, const float* scale, const float* bias,
                          int64_t M, int K, int N, int act, const float* residual, int ldr, const int* rowmap,
                          float* out, int ldc, int out_pair, void* stream);
int al3d_tok_window_attention_f32(const float* qkv, const float* table, int nwin, int C, int heads, int win_rows,
                                  int win_cols, int shift, float scale, int out_pair, float* out, void* stream);
int al3d_tok_window_attention_tokens_f32(const float* qkv, const float* bias_qkv, const float* table, int B, int H, int W,
                                         int C, int heads, int shift, float scale, int out_pair, float* out, void* stream);
/* Multi-head attention with 16-channel heads, any number of keys (the TransFusion query decoder,
 * bevfusion/mmdet3d/models/utils/transformer.py:71-112 -> nn.MultiheadAttention's core: softmax(q scale k^T) v after
 * the input projections, before the output projection).  q [B][Pq][ldq], k [B][Pk][ldk], v [B][Pk][ldv] f32 with head
 * h in columns h*16 .. h*16+15; out [B][Pq][ldo].  Keys are split into chunks of <= 1,024 handled by independent
 * waves with an online softmax; workspace >= al3d_tok_mha16_workspace_bytes(). */
int64_t al3d_tok_mha16_workspace_bytes(int B, int heads, int Pq, int Pk);
int al3d_tok_mha16_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int B, int heads, int Pq,
                       int Pk, float scale, float* out, int ldo, void* workspace, void* stream);

/* ---------------------------------------------------------------- runtime
 * A HIP stream restricted to n_cus compute units starting at first_cu (hipExtStreamCreateWithCUMask); the
 * reference has no analogue (its loader workers are host processes, det3d/datasets/loader/build_loader.py:23-59):
 * here the next batch's voxelizer + rulebook and the decode/NMS kernels run on such a stream next to the main
 * stream's convolutions.  The stream lives until process exit. */
int al3d_stream_create_cu_mask(int n_cus, int first_cu, void** out_stream);

#ifdef __cplusplus
}
#endif
#endif /* AL3D_H_ */
