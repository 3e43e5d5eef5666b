/*
 * tmae_hip.h -- C ABI of libtmae_hip.so: the MI355X (gfx950) operators of the T-MAE
 * pre-training hot path.  This is the drop-in boundary (level B4): these
 * entry points replace what the reference reaches through
 *   - its in-tree pybind module  sst_ops_cuda   (pcdet/ops/sst_ops/src/sst_ops_api.cpp:6-9)
 *   - torch_scatter               (pcdet/models/backbones_3d/vfe/temporal_dyn_vfe.py:85,113)
 *   - torch.unique(dim=0)         (temporal_dyn_vfe.py:72)
 *   - spconv.pytorch              (pcdet/utils/spconv_utils.py:37-56, SiamWCA_MAE.py:187-193,235)
 *   - pytorch3d.loss              (pcdet/models/backbones_3d/SiamWCA_MAE.py:163)
 *   - the padded torch.bmm attention of cosine_msa.py:114-176 + sst_utils.py:118-192
 *
 * Conventions (all functions):
 *   - plain C: device pointers, sizes, one hipStream_t (passed as void*); no torch types.
 *   - the CALLER owns every buffer (outputs and workspace); nothing is allocated,
 *     freed or synchronised inside; kernels are enqueued on `stream` and the call returns.
 *   - return 0 on success, <0 for a bad argument (TMAE_E*), >0 = hipError_t of the launch.
 *   - `dtype`: 0 = float32, 1 = bfloat16 for feature tensors; index tensors are int32
 *     unless the name says i64 (the int64 ones mirror the reference's batch_dict keys).
 *   - thread-safe for distinct streams + distinct workspaces.
 */
#ifndef TMAE_HIP_H
#define TMAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMAE_OK 0
#define TMAE_EARG (-1)   /* null pointer / negative size / unsupported shape */
#define TMAE_EWS (-2)    /* workspace too small */
#define TMAE_EDTYPE (-3) /* unsupported dtype */

#define TMAE_F32 0
#define TMAE_BF16 1

/* Library identification.  TMAE_ABI_VERSION is bumped by hand on every change of the export list, of a signature or of what an
 * entry point does with its arguments; tmae_abi_hash() is the fingerprint of THIS header's prototypes (name, return class and
 * argument classes in order: tmae_amd/_abi.py) that the build compiled in.  The Python binding compares both at import, so a
 * stale library, or a binding row that disagrees with its prototype, fails there and not inside a launch. */
#define TMAE_ABI_VERSION 25
int tmae_abi_version(void);
int tmae_abi_hash(void);

/* ---------------------------------------------------------------------------------------
 * A1  dynamic voxelisation.   Replaces get_in_range_mask (pcdet/utils/common_utils.py:66-76)
 * + boolean compaction + torch.unique(dim=0, return_inverse) (temporal_dyn_vfe.py:67-72).
 * points [n,row] f32 rows (b,x,y,z,features...), row = 1 + point features (5 for ONCE, 6 for Waymo).
 * Outputs sized for the worst case n / n_cells:
 *   points_out [n,row], point_coords_i64 [n,4] (b,z,y,x), inverse_i64 [n], voxel_coords_i64 [n,4]
 *   (lexicographically sorted (b,z,y,x), exactly what unique(dim=0) yields),
 *   counts [2+batch] device int32: {n_kept, n_voxels, voxels of sample 0, 1, ...}.
 * Coordinates use IEEE fp32 subtract + divide and truncation toward zero (bit-exact). */
size_t tmae_voxelize_workspace(int64_t n, int batch, int gx, int gy, int gz);
int tmae_voxelize(const float* points, int row, int64_t n, int batch,
                  float rmin_x, float rmin_y, float rmin_z, float vs_x, float vs_y, float vs_z,
                  int gx, int gy, int gz,
                  float* points_out, int64_t* point_coords_i64, int64_t* inverse_i64,
                  int64_t* voxel_coords_i64, int32_t* counts,
                  void* ws, size_t ws_bytes, void* stream);

/* CSR of points per voxel in ascending point order (stable): perm [n] point ids grouped by
 * voxel, offsets [m+1].  Canonical form of the atomic order in sst_ops_gpu.cu:14-28. */
size_t tmae_segment_csr_workspace(int64_t n, int64_t m);
int tmae_segment_csr(const int64_t* inverse_i64, int64_t n, int64_t m,
                     int32_t* perm, int32_t* offsets, void* ws, size_t ws_bytes, void* stream);

/* A5  sst_ops.get_inner_win_inds (pcdet/ops/sst_ops/sst_ops_utils.py:5-12 ->
 * sst_ops_gpu.cu:14-20): running index of each element inside its group; canonical =
 * stable rank in element order.  group ids in [0, num_groups). */
size_t tmae_ingroup_rank_workspace(int64_t n, int64_t num_groups);
int tmae_ingroup_rank(const int64_t* group_i64, int64_t n, int64_t num_groups, int64_t* out_i64,
                      void* ws, size_t ws_bytes, void* stream);

/* A2  fused voxel mean (torch_scatter.scatter mean, temporal_dyn_vfe.py:85) + the F+6 point
 * features [f_center(3) | x y z i ... (F = row-1) | f_cluster(3)] (temporal_dyn_vfe.py:88-109).
 * points [n,row] (kept points), point_coords_i64 [n,4], inverse [n]; CSR from tmae_segment_csr.
 * Outputs voxel_mean [m,F] f32, feats [n,F+6] f32. */
int tmae_vfe_point_features(const float* points, int row, const int64_t* point_coords_i64,
                            const int64_t* inverse_i64, const int32_t* perm, const int32_t* offsets,
                            int64_t n, int64_t m,
                            float rmin_x, float rmin_y, float rmin_z, float vs_x, float vs_y, float vs_z,
                            float* voxel_mean, float* feats, void* stream);

/* The same point features for the bf16 (autocast) path, which must not round absolute coordinates (up to 75 m) to
 * 8 mantissa bits before the first Linear (temporal_dyn_vfe.py:110-112 under AMP keeps 11): feats_hl [n,32] bf16 =
 * [hi(16) | lo(16)] with hi + lo = the fp32 feature to ~2^-17 relative, columns >= F+6 of each half zero.
 * inverse_csr (may be NULL) [n] i64: when given, the rows of feats_hl come out in CSR order -- row j = point perm[j], i.e.
 * sorted by voxel -- and inverse_csr[j] = the voxel of that row: the point MLP is row-wise, and the segment max behind it
 * (tmae_segment_max_fwd with perm = NULL, _bwd with inverse_csr) then walks consecutive rows. */
int tmae_vfe_point_features_bf16x2(const float* points, int row, const int64_t* point_coords_i64,
                                   const int64_t* inverse_i64, const int32_t* perm, const int32_t* offsets,
                                   int64_t n, int64_t m,
                                   float rmin_x, float rmin_y, float rmin_z, float vs_x, float vs_y, float vs_z,
                                   float* voxel_mean, void* feats_hl, int64_t* inverse_csr, void* stream);

/* torch_scatter.scatter_max (temporal_dyn_vfe.py:113): out[v,c] = max over the voxel's points,
 * argmax[v,c] = first point (ascending id) attaining it.  x [n,c]; backward routes the
 * gradient to the argmax rows (dx fully written, no pre-zeroing needed).  perm may be NULL: the rows of x are in CSR
 * order already (row j belongs to the voxel v with offsets[v] <= j < offsets[v+1]); argmax then holds CSR row ids. */
int tmae_segment_max_fwd(const void* x, int dtype, int64_t n, int64_t m, int c,
                         const int32_t* perm, const int32_t* offsets,
                         void* out, int32_t* argmax, void* stream);
/* tmae_segment_max_fwd over the rows of relu?(BatchNorm(x)) without writing them: the VFE's last Linear -> BatchNorm1d -> ReLU ->
 * scatter_max (temporal_dyn_vfe.py:110-113) with the norm's statistics given (tmae_bn_stats).  out / argmax are bit for bit those
 * of tmae_bn_apply followed by tmae_segment_max_fwd.  c in {64, 128, 256}; the backward is tmae_segment_max_bwd then
 * tmae_bn_relu_bwd. */
int tmae_segment_max_bn_fwd(const void* x, int dtype, int64_t n, int64_t m, int c, const int32_t* perm, const int32_t* offsets,
                            const float* mean, const float* rstd, const float* gamma, const float* beta, int relu, void* out,
                            int32_t* argmax, void* stream);
int tmae_segment_max_bwd(const void* dout, int dtype, int64_t n, int64_t m, int c,
                         const int64_t* inverse_i64, const int32_t* argmax, void* dx, void* stream);

/* A12  sst_ops.group_inner_inds (sst_ops_utils.py:15-27 -> sst_ops_gpu.cu:22-39) fused with the
 * target normalisation of SiamWCA_MAE.target_assigner (SiamWCA_MAE.py:134-141):
 * group_inds_i64 [m,k] = first k point ids per voxel (point order), cyclic repeat when fewer;
 * gt [m,k,3] = xyz[group_inds] - voxel centre ((coord+0.5)*vs+rmin, common_utils.py:130-145). */
int tmae_group_points(const float* points, int row, const int64_t* voxel_coords_i64,
                      const int32_t* perm, const int32_t* offsets, int64_t m, int k,
                      float rmin_x, float rmin_y, float rmin_z, float vs_x, float vs_y, float vs_z,
                      int64_t* group_inds_i64, float* gt, void* stream);

/* A3  random masking with injected noise (common_utils.py:49-63, SiamWCA_MAE.py:166-182).
 * noise [m] f32 >= 0 in voxel order, sample_offsets [batch+1] device int32 (voxel rows are
 * grouped by sample).  keep the int(L*keep_frac) smallest-noise voxels of each sample
 * (ties by lower index, = stable argsort).  mask [m] f32 (1 = removed),
 * vis_index [m] int32: compacted ids of kept voxels (first n_vis entries), n_vis device int32. */
size_t tmae_random_mask_workspace(int64_t m, int batch);
int tmae_random_mask(const float* noise, const int32_t* sample_offsets, int64_t m, int batch,
                     double keep_frac, float* mask, int32_t* vis_index, int32_t* n_vis,
                     void* ws, size_t ws_bytes, void* stream);

/* Dense index grid of a sparse tensor: grid [batch*ny*nx] int32 = row id or -1.
 * indices [m,3] int32 (b,y,x) (the spconv SparseConvTensor.indices layout, SiamWCA_MAE.py:187-193). */
int tmae_index_grid(const int32_t* indices, int64_t m, int batch, int ny, int nx,
                    int32_t* grid, void* stream);

/* A4/A10  window partition + region batching (SSTInputLayer.forward spt_backbone.py:137-184;
 * get_window_coors sst_utils.py:6-58; drop_single_shift spt_backbone.py:47-71;
 * get_flat2win_inds sst_utils.py:79-107; joint two-frame variant
 * SSTInputLayer_Temporal.drop_single_shift_ref_to_prv SiamWCA.py:65-140).
 * grid = dense index grid of this frame, grid_other = the other frame's grid or NULL
 * (single-frame).  do_shift 0: windows offset by a full window, 1: by half (sst_utils.py:29-36).
 * levels: n_levels rows {max_tokens, lower, upper}.  Per-voxel outputs [m]:
 *   batch_win_inds_i64, coors_in_win_i64 [m,3] (z,y,x), inner_i32 (stable rank in window),
 *   level_i32, keep_u8, flat2win_i64 (= rank_of_window_in_level*max_tokens + inner; -1 if dropped).
 *   win_per_level [n_levels] device int32 = number of kept windows per level.
 * batch_win_inds / coors_in_win / level / keep / flat2win may each be NULL (not wanted).  flat2win and win_per_level both NULL:
 * the per-level window ranks (one device scan per level) are not computed at all -- the cross-attention blocks, which need
 * keep_u8 alone (wca_block.py:93-96), run three launches instead of 3 + n_levels. */
size_t tmae_window_bucket_workspace(int batch, int ny, int nx, int wy, int wx, int n_levels);
int tmae_window_bucket(const int32_t* indices, int64_t m, const int32_t* grid, const int32_t* grid_other,
                       int batch, int ny, int nx, int wy, int wx, int do_shift,
                       const int32_t* levels_host, int n_levels,
                       int64_t* batch_win_inds_i64, int64_t* coors_in_win_i64, int32_t* inner_i32,
                       int32_t* level_i32, uint8_t* keep_u8, int64_t* flat2win_i64,
                       int32_t* win_per_level, void* ws, size_t ws_bytes, void* stream);

/* A6+A7  ragged window cosine attention (replaces flat2window/window2flat sst_utils.py:118-192 and
 * _scaled_cosine_attention cosine_msa.py:114-176).  One workgroup per (window, head group); the
 * window's tokens are read straight from the dense index grids, so there is no padding and no
 * per-level loop.  q [mq, ldq], k/v [mk, ldk/ldv] already projected (heads contiguous: head h =
 * columns [h*dh, (h+1)*dh)); per-head L2 normalisation (eps 1e-12), logits / max(tau, tau_min),
 * softmax over the window's keys, P.V.  Self-attention: grid_k == grid_q.  Cross-attention
 * (wca_block.py:26-67): grid_q = current frame, grid_k = previous frame; query tokens whose
 * window has no key get a zero row (they are not "kept", wca_block.py:93-96).
 * lse [mq, nhead] f32 is saved for the backward.  dh in {16, 32}; 8x8 windows.
 * tau_per_head 0: tau[0] is the layer's one temperature (cosine_msa.py:455-456); 1: tau [nhead], one per head
 * (non_shared_tau, cosine_msa.py:453-454, :155-158). */
int tmae_win_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      int dtype, int64_t mq, int64_t mk, int nhead, int dh,
                      const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx,
                      int do_shift, const float* tau, float tau_min,
                      void* out, int64_t ldo, float* lse, const int32_t* worklist, int tau_per_head, void* stream);
/* Backward.  dq/dk/dv are fully written for every token that sits in an attended window and
 * zero-filled otherwise.  dtau_partial [n_windows*nhead/heads_per_block] f32 partial sums of
 * d loss / d max(tau,tau_min) (summed by the caller; n entries = tmae_win_attn_num_blocks). */
int64_t tmae_win_attn_num_blocks(int batch, int ny, int nx, int nhead, int dh);
/* dtau[0] = tau >= tau_min ? -(sum of the n partials) / tau : 0  -- the chain rule through
 * clamp(tau, min=tau_min) of cosine_msa.py:150-152; fixed-order sum over up to 64 strips, one launch.  Calls must be
 * ordered on one stream (a module-scope ticket counter picks the block that finishes).  worklist (may be NULL): the work list the
 * backward ran with -- windows that are in none of its lists wrote no partial and count as zero (no pre-zeroing of dtau_partial);
 * NULL: every entry is read.  tau_per_head 1: tau / dtau [nhead], dtau[h] from the partials of head h (n = windows x nhead,
 * one launch per head). */
int tmae_win_attn_dtau(const float* dtau_partial, int64_t n, const float* tau, float tau_min, float* dtau,
                       const int32_t* worklist, int nhead, int tau_per_head, void* stream);
int tmae_win_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                      int dtype, int64_t mq, int64_t mk, int nhead, int dh,
                      const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx,
                      int do_shift, const float* tau, float tau_min,
                      void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                      float* dtau_partial, const int32_t* worklist, int tau_per_head, void* stream);

/* Window work lists -- the reference's region batching (drop levels 16/32/64 tokens, spt_backbone.py:47-71,
 * t_mae_ssl.yaml:63-67) as three index lists instead of padded tensors: windows of one shift that hold both queries
 * and keys, binned by the 16-token tiles they need.  worklist [tmae_window_worklist_size()] int32:
 * [0..2] counts, [4 + c*nwin + j] dense window ids of class c.  Passed to tmae_win_attn_fwd/bwd (bf16 path) it
 * selects kernel instantiations sized for the class; tokens of windows that are in no list (no key / no query in the
 * other frame) are then NOT written: the caller pre-zeroes out / dq / dk / dv and dtau_partial.  NULL = one
 * worst-case kernel over all dense windows. */
/* The complement of the work lists for cross attention: zeroes the rows that no list-driven launch writes -- rows of q0 [., wq0]
 * (bf16, pitch ldq0) and qf [., nqf] (f32) of the query tokens whose window holds no key, rows of k0 / k1 [., wk] of the key tokens
 * whose window holds no query (any pointer may be NULL) -- instead of pre-zeroing whole tensors.  Valid when every token of the two
 * grids' token lists lies in a window (no token dropping).  Widths and pitches multiples of 8 elements, 16-byte aligned bases. */
int tmae_win_attn_zero_orphans(const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx, int do_shift, void* q0,
                               int64_t ldq0, int wq0, float* qf, int nqf, void* k0, int64_t ldk0, void* k1, int64_t ldk1, int wk,
                               void* stream);
size_t tmae_window_worklist_size(int batch, int ny, int nx);
int tmae_window_worklist(const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx, int do_shift,
                         int32_t* worklist, void* stream);

/* SSTInputLayer.get_pos_embed (spt_backbone.py:186-224) fused with the q = k = x + pos add of
 * WindowAttention.forward (sst_basic_block.py:41-44): out[r,:] = x[r,:] + pos_table[cell(r),:], where
 * cell(r) = ((y+sy)%wy)*wx + (x+sx)%wx is the token's position inside its (shifted) window and
 * pos_table [wy*wx, d] f32 holds the sin/cos embedding of every in-window position (built once per
 * width by the host from the reference formula).  indices [m,3] int32 (b,y,x). */
int tmae_add_pos_embed(const void* x, int dtype, int64_t m, int d, const int32_t* indices,
                       int wy, int wx, int do_shift, const float* pos_table, void* out, void* stream);

/* A9  2-D sparse convolution rulebook (spconv SparseConv2d k3 s2 p1 / SubMConv2d k3;
 * spt_backbone.py:280-304).
 * down_outputs: active output sites of the strided conv, lexicographic (b,y,x):
 *   out_grid [batch*oy*ox] int32 row ids (-1 inactive), out_indices [<=batch*oy*ox,3], n_out device. */
size_t tmae_spconv_down_outputs_workspace(int batch, int oy, int ox);
int tmae_spconv_down_outputs(const int32_t* grid_in, int batch, int ny, int nx, int oy, int ox,
                             int32_t* out_grid, int32_t* out_indices, int32_t* n_out,
                             void* ws, size_t ws_bytes, void* stream);
/* neighbour tables.  nbr [m_out,9]: input row read by output o at tap t=ky*3+kx (or -1):
 * input site = (oy*stride-1+ky, ox*stride-1+kx).  nbr_t [m_in,9]: output row that reads input i
 * at tap t (or -1) -- the transposed rulebook used by the data gradient. */
int tmae_spconv_neighbors(const int32_t* out_indices, int64_t m_out, const int32_t* grid_in,
                          int batch, int ny, int nx, int stride, int32_t* nbr, void* stream);
int tmae_spconv_neighbors_t(const int32_t* in_indices, int64_t m_in, const int32_t* grid_out,
                            int batch, int oy, int ox, int stride, int32_t* nbr_t, void* stream);
/* A9  the sparse convolution itself (spconv SparseConv2d / SubMConv2d forward and input gradient,
 * pcdet/utils/spconv_utils.py:37-56, spt_backbone.py:280-304) as an implicit GEMM over the rulebook: bf16 in / out,
 * fp32 accumulation, no [m, 9 cin] im2col matrix.
 *   fwd:      out [m_out, cout] = sum_t feat[nbr[o,t], :] . weight[:, t, :]^T      weight [cout, 9*cin] (spconv-2 layout
 *             [cout,3,3,cin] flattened), nbr [m_out, 9] from tmae_spconv_neighbors (-1 = no neighbour)
 *   bwd_data: din [m_in, cin] = sum_t dout[nbr_t[i,t], :] . weight_t[:, t, :]^T    weight_t [cin, 9*cout] with
 *             weight_t[c, t*cout + n] = weight[n, t*cin + c], nbr_t from tmae_spconv_neighbors_t
 * cin, cout in {128, 256, 384} (384: the dense decoder's concatenated input, SiamWCA_MAE.py:100-115, run through
 * the rulebook of a full grid); row pitches ld* in elements, multiples of 8; 16-byte aligned bases. */
int tmae_spconv_fwd(const void* feat, int64_t ldf, int64_t m_in, int cin, const int32_t* nbr, int64_t m_out,
                    const void* weight, int cout, void* out, int64_t ldo, void* stream);
int tmae_spconv_bwd_data(const void* dout, int64_t lddo, int64_t m_out, int cout, const int32_t* nbr_t, int64_t m_in,
                         const void* weight_t, int cin, void* din, int64_t lddi, void* stream);

/* A11  the dense decoder's Conv2d(cin, cout, 3, padding=1, bias=False) (SiamWCA_MAE.py:100-115) on a channels-last
 * grid [batch, ny, nx, cin] bf16 -> [batch, ny, nx, cout] bf16, fp32 accumulation: halo-tiled implicit GEMM (a workgroup
 * stages the 18x18 input halo of its 16x16 output block once per 64-channel slice).  weight [cout, 9*cin]: the
 * [cout,3,3,cin] layout flattened.  cin in {128,256,384}, cout % 128 == 0.  Input gradient = the same call on dout with
 * weight_t[c, ((2-ky)*3 + (2-kx))*cout + n] = weight[n, (ky*3+kx)*cin + c]. */
int tmae_dense_conv3x3(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout, void* out,
                       void* stream);
/* The same convolution with dilation d in {1, 2} and padding d (the dilated conv of SSTBEVBackbone,
 * sst_bev_backbone.py:20-30, t_mae.yaml:107-112): 16 x 16 cell blocks with a (16 + 2d)^2 halo. */
int tmae_dense_conv3x3_dilated(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout,
                               int dilation, void* out, void* stream);
/* out = conv + post (post: the shape of out, bf16): the input gradient of a residual block's conv plus the gradient that
 * arrives through its shortcut (autograd's gradient accumulation at `out` in sst_bev_backbone.py:35-41), one rounding. */
int tmae_dense_conv3x3_add(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout, int dilation,
                           const void* post, void* out, void* stream);
/* The decoder conv's two heavy launches with the column sums of their output taken in the epilogue (sums of the bf16 values the
 * kernel stores; fixed summation order): moments = 1 (cin 128, cout 384: the input gradient of the conv, summed by the BatchNorm
 * backward of the three deconvolutions in front of it, SiamWCA_MAE.py:85-99) -> sums [cout]; moments = 2 (cin 384, cout 128: the
 * forward, SiamWCA_MAE.py:100-115, whose BatchNorm2d needs mean and variance) -> sums [2][cout] = sum, sum of squares.  post:
 * optional residual operand as in tmae_dense_conv3x3_add.  Any other shape: TMAE_EARG. */
size_t tmae_dense_conv3x3_sums_workspace(int cout);
int tmae_dense_conv3x3_sums(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout, const void* post,
                            int moments, void* out, float* sums, void* ws, size_t ws_bytes, void* stream);
/* Weight gradient of that convolution (SiamWCA_MAE.py:100-115, sst_bev_backbone.py:20-30; torch's conv2d weight gradient in
 * the reference): dw [cout, 9*cin] fp32 (the weight's layout) from dy [batch, ny, nx, cout] and x [batch, ny, nx, cin], bf16,
 * contiguous.  Halo-tiled like the forward (the nine shifted copies of x are read out of one staged image), fixed-order
 * slab reduction.  cout = 128, cin % 64 == 0 (<= 512), dilation in {1, 2}; TMAE_EARG otherwise (then: tmae_spconv_wgrad
 * over a full-grid rulebook).  ws: tmae_dense_conv3x3_wgrad_workspace(cin, cout) bytes. */
size_t tmae_dense_conv3x3_wgrad_workspace(int cin, int cout);
int tmae_dense_conv3x3_wgrad(const void* dy, const void* x, int batch, int ny, int nx, int cin, int cout, int dilation,
                             float* dw, void* ws, size_t ws_bytes, void* stream);

/* 3x3 convolutions (padding 1) between a 64-channel channels-last map and a narrow one (1 <= k <= 8 channels): the last convs of
 * CenterHead's branches (pcdet/models/dense_heads/center_head.py:11-45; torch.nn.Conv2d(64, k, 3, padding=1, bias=True) forward,
 * its input gradient and its weight gradient in the reference).  in / din [batch, ny, nx, 64] bf16 with `ld` elements per cell
 * (ld >= 64, ld % 8 == 0: a 64-channel slice of a wider map may be passed), out / dout [batch, ny, nx, k] bf16 contiguous,
 * weight [k, 9, 64] bf16 (the [k, 3, 3, 64] layout flattened), bias [k] fp32 (added before the one rounding to bf16),
 * dw [k, 9, 64] fp32.  Every tensor < 2 GB.  ws: the matching *_workspace() bytes, 16-byte aligned. */
int tmae_conv3x3_c64_narrow_fwd(const void* in, int64_t ldi, int batch, int ny, int nx, const void* weight, const float* bias,
                                int k, void* out, void* stream);
size_t tmae_conv3x3_c64_narrow_bwd_data_workspace(void);
int tmae_conv3x3_c64_narrow_bwd_data(const void* dout, int batch, int ny, int nx, int k, const void* weight, void* din,
                                     int64_t ldo, void* ws, size_t ws_bytes, void* stream);
size_t tmae_conv3x3_c64_narrow_wgrad_workspace(int k);
int tmae_conv3x3_c64_narrow_wgrad(const void* dout, const void* in, int64_t ldi, int batch, int ny, int nx, int k, float* dw,
                                  void* ws, size_t ws_bytes, void* stream);

/* 3x3 convolution (padding 1) 64 -> 64 channels on a channels-last bf16 map -- the stem convs of CenterHead's branches
 * (center_head.py:28-31: torch.nn.Conv2d(64, 64, 3, padding=1, bias=False) in front of BatchNorm + ReLU) -- with weight
 * [64, 9, 64] bf16 (the [64, 3, 3, 64] layout flattened).  input_gradient = 0: out[.., n] = sum in[.. + tap, c] weight[n, tap, c];
 * input_gradient = 1: `in` is the output gradient, `out` the input gradient (taps flipped, weight transposed inside).  ldi / ldo /
 * lddo: elements per cell (>= 64, % 8 == 0).  accumulate = 1: out += (the second 64-channel half of a wider contraction: the shared
 * conv 128 -> 64 of CenterHead, center_head.py:85-89, is two calls on the halves of its input).  dw [64, 9, 64] fp32.  ws: the matching *_workspace() bytes, 16-byte aligned. */
size_t tmae_conv3x3_c64_workspace(void);
int tmae_conv3x3_c64(const void* in, int64_t ldi, int batch, int ny, int nx, const void* weight, int input_gradient, int accumulate,
                     void* out, int64_t ldo, void* ws, size_t ws_bytes, void* stream);
size_t tmae_conv3x3_c64_wgrad_workspace(void);
int tmae_conv3x3_c64_wgrad(const void* dout, int64_t lddo, const void* in, int64_t ldi, int batch, int ny, int nx, float* dw,
                           void* ws, size_t ws_bytes, void* stream);

/* gather-GEMM form: cols [m_out, 9*c] = rows of feat selected by nbr (zeros where -1), to be
 * multiplied by the [cout, 9*c] view of the spconv-2 weight [cout,3,3,cin]; and its adjoint
 * din [m_in, c] = sum_t dcols[nbr_t[i,t], t*c:(t+1)*c]. */
int tmae_spconv_gather(const void* feat, int dtype, int64_t m_in, int c, const int32_t* nbr,
                       int64_t m_out, void* cols, void* stream);
int tmae_spconv_gather_t(const void* dcols, int dtype, int64_t m_out, int c, const int32_t* nbr_t,
                         int64_t m_in, void* din, void* stream);

/* SparseConvTensor.dense() (SiamWCA_MAE.py:235) in channels-last: out [batch,ny,nx,c], zeros at
 * inactive sites; and its adjoint / the decoder-feature gather (SiamWCA_MAE.py:308-312):
 * rows [m,c] = dense[b,y,x,:]. */
int tmae_sparse_to_dense(const void* feat, int dtype, int64_t m, int c, const int32_t* grid,
                         int batch, int ny, int nx, void* out, void* stream);
int tmae_dense_gather(const void* dense, int dtype, int batch, int ny, int nx, int c,
                      const int32_t* indices, int64_t m, void* rows, void* stream);

/* A13  pytorch3d.loss.chamfer_distance(pred, gt, weights) (SiamWCA_MAE.py:154-164; v0.7.1 defaults):
 * per voxel: cx = mean_i min_j |p_i-g_j|^2, cy = mean_j min_i |g_j-p_i|^2.  One wavefront per voxel,
 * one gt point per lane (ng <= 64, np <= 64).  per_voxel [m] = w*(cx+cy); idx_x [m,np] / idx_y [m,ng]
 * int8 nearest ids saved for the backward.  loss = sum(per_voxel)/sum(w) is finished by the caller. */
int tmae_chamfer_fwd(const float* pred, const float* gt, const float* weights, int64_t m, int np, int ng,
                     float* per_voxel, int8_t* idx_x, int8_t* idx_y, void* stream);
/* dpred [m,np,3] = scale[0] * w * (2/np (p_i - g_nn(i)) + 2/ng sum_{j: nn(j)=i} (p_i - g_j)). */
int tmae_chamfer_bwd(const float* pred, const float* gt, const float* weights, const int8_t* idx_x,
                     const int8_t* idx_y, const float* scale, int64_t m, int np, int ng,
                     float* dpred, void* stream);

/* Weight / bias gradient of the token-list Linear layers (replaces the autograd of torch.nn.functional.linear as
 * called from cosine_msa.py:47-62,431, sst_basic_block.py:81, wca_block.py:99 and of the gather-GEMM sparse conv):
 *   dw [n,k] f32 = sum_m dy[m,n] * x[m,k];   db [n] f32 = sum_m dy[m,n]  (db may be NULL).
 * dy [m, >=n] and x [m, >=k] are bf16 with row strides ldy / ldx (elements, multiples of 8, 16-byte aligned bases);
 * n, k multiples of 8.  Token axis split over ~1k workgroups, MFMA 16x16x32 with transposed LDS reads, fp32 slabs
 * in the workspace summed in a fixed order (deterministic). */
size_t tmae_linear_wgrad_workspace(int64_t m, int n, int k);
int tmae_linear_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t m, int n, int k,
                      float* dw, float* db, void* ws, size_t ws_bytes, void* stream);
/* The same pass for the attention in-projections with the position embedding folded into the GEMM
 * (tmae_token_gemm_pos): besides dw / db it returns dcell [16, n] f32, the per-cell column sums of dy --
 * dcell[c, j] = sum of dy[i, j] over the tokens i with xc == c (c < 8) / yc == c - 8 (c >= 8), for the output columns
 * j < pos_n (the others are not defined) -- so that dW[:pos_n] += dcell[:, :pos_n]^T . E (E [16,k]: the separable embedding)
 * completes the gradient of (x + pos) W^T (ONE pass over dy for both tile sizes: the one-hot products ride on the
 * weight-gradient kernel's own dy fragments).  pos_e (may be NULL): E [16,k] f32, 16-byte aligned -- the slab reduction then
 * adds dcell^T . E to dw itself (needs k in {64, 128, 256, 512, ...}: 256 % k == 0 or k % 256 == 0, and n k % 256 == 0;
 * TMAE_EARG otherwise) and dcell may be NULL.  cells: u8 xc | yc << 3 per token, 8-byte aligned, readable up to
 * round_up(m, 32) + 64 bytes (tmae_window_cells callers pad the buffer).  Same workspace. */
int tmae_linear_wgrad_cells(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t m, int n, int k,
                            const uint8_t* cells, int pos_n, const float* pos_e, float* dw, float* db, float* dcell,
                            void* ws, size_t ws_bytes, void* stream);

/* Weight gradient of the 3x3 sparse conv without materialising the gathered [m_out, 9*cin] matrix: the token-split
 * kernel reads row nbr[o,t] of feat [m_in, cin] (bf16) for the column block of tap t.  dw [cout, 9*cin] f32 = the
 * spconv-2 weight layout [cout,3,3,cin] flattened; cin a multiple of 128, cout of 8; workspace =
 * tmae_linear_wgrad_workspace(m_out, cout, 9*cin). */
int tmae_spconv_wgrad(const void* dy, int64_t ldy, const void* feat, int64_t ldf, const int32_t* nbr, int64_t m_out,
                      int cout, int cin, float* dw, void* ws, size_t ws_bytes, void* stream);

/* Fused residual add + LayerNorm of the post-norm encoder layers (EncoderLayer.forward sst_basic_block.py:77-84,
 * wca_block.py:93-102): y = LN(a + b) * gamma + beta, eps inside the sqrt, biased variance (nn.LayerNorm).
 * a, b (b may be NULL), y, xsum [m,d] in `dtype`; d in {128, 256}; xsum (may be NULL) receives a + b for the
 * backward; mean / rstd [m] f32.  Statistics are taken in fp32 over the stored (rounded) sum.
 * bmask (may be NULL; [m] in `dtype`): row r of b is scaled by bmask[r] -- `src[keep_inds] += attn` of the cross layers
 * (wca_block.py:93-96) as a 0/1 row weight.  post (may be NULL; [m,d] in `dtype`): y = LN(...) + post -- the block residual
 * `x + encoder(x)` of SSTBlockV1.forward / WCABlock.forward (spt_backbone.py:342-353, SiamWCA.py:431-447) on the
 * block's last norm. */
int tmae_add_layernorm_fwd(const void* a, const void* b, int dtype, int64_t m, int d, const float* gamma,
                           const float* beta, float eps, void* xsum, void* y, float* mean, float* rstd,
                           const void* bmask, const void* post, void* stream);
/* dx [m,d] = gradient wrt (a + b); dgamma / dbeta [d] f32 (two-stage fixed-order column sums).  Optional pairs (both NULL or
 * both given): skip / dx_skip: dx_skip = dx + skip (a gradient that reaches the first summand by another path, summed
 * here instead of autograd's add); bmask / dx_b: dx_b = dx * bmask[row] (the gradient of the masked second summand).  dx
 * itself may be NULL when nobody reads it. */
size_t tmae_layernorm_bwd_workspace(int64_t m, int d);
int tmae_layernorm_bwd(const void* dy, const void* x, int dtype, int64_t m, int d, const float* mean,
                       const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta,
                       const void* skip, void* dx_skip, const void* bmask, void* dx_b,
                       void* ws, size_t ws_bytes, void* stream);

/* BatchNorm1d with batch statistics (+ optional ReLU) over the rows of a token / point list -- the norm of
 * spconv_utils.post_act_block (pcdet/utils/spconv_utils.py:50-54, eps 1e-3) and of the VFE MLP
 * (model_utils/network_utils.py:31).  x, y, dy, dx [m,c] in `dtype`, c in {64,128,256}; mean / var (biased) / rstd [c]
 * f32 are outputs of the forward (the caller updates the running statistics from mean and var*m/(m-1)).
 * y = relu?((x - mean) * rstd * gamma + beta).  Backward: dx, dgamma, dbeta; the ReLU mask is recomputed from x. */
size_t tmae_bn_workspace(int64_t m, int c);
int tmae_bn_relu_fwd(const void* x, int dtype, int64_t m, int c, const float* gamma, const float* beta, float eps,
                     int relu, void* y, float* mean, float* var, float* rstd, void* ws, size_t ws_bytes, void* stream);
/* The same forward with a residual operand: y = relu?(norm(x)) + post ([m,c] in `dtype`) -- `out = conv_bn_relu(out) + out` of
 * SSTBEVBackbone (sst_bev_backbone.py:35-41): the shortcut is added where the normalised row is in registers.  The backward is
 * tmae_bn_relu_bwd (the ReLU mask is recomputed from x), the gradient of post is dy itself. */
int tmae_bn_relu_add_fwd(const void* x, int dtype, int64_t m, int c, const float* gamma, const float* beta, float eps,
                         int relu, const void* post, void* y, float* mean, float* var, float* rstd, void* ws,
                         size_t ws_bytes, void* stream);
int tmae_bn_relu_bwd(const void* dy, const void* x, int dtype, int64_t m, int c, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int relu, void* dx, float* dgamma, float* dbeta,
                     void* ws, size_t ws_bytes, void* stream);
/* The same backward for a gradient that arrives in two pieces, dy + dy2 (both [m,c] in `dtype`, summed in fp32 inside the two
 * passes): the output of a stage's last BatchNorm feeds the next stage AND, split by frame, the cross-attention block
 * (SiamWCA_MAE.py:262-291); the caller hands over both gradients instead of adding them first. */
int tmae_bn_relu_bwd2(const void* dy, const void* dy2, const void* x, int dtype, int64_t m, int c, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int relu, void* dx, float* dgamma,
                      float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* Backward of the same norm over the rows of a DENSE map x [batch, ny, nx, c] whose output is read at m sites only -- the decoder's
 * last BatchNorm2d + ReLU, read back by the gather at the current frame's voxels (SiamWCA_MAE.py:100-115, 303-312).  dyc [m, c]: the
 * gradient at site indices[j] = (b, y, x) (int32 [m, 3], unique); rowmap [batch * ny * nx] int32: the site's row in dyc or -1
 * (tmae_index_grid of `indices`); all other cells carry zero gradient.  The two sums run over the m rows, dx [batch * ny * nx, c]
 * is written for every cell; workspace: tmae_bn_workspace(m, c). */
int tmae_bn_relu_bwd_gathered(const void* dyc, const int32_t* indices, const int32_t* rowmap, int64_t m, const void* x, int dtype,
                              int batch, int ny, int nx, int c, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, int relu, void* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                              void* stream);

/* Split BatchNorm entry points with an explicit element count (used by the fused decoder head below): statistics
 * of the m stored rows normalised by `count` >= m (the rows that are not stored are exact zeros), the two backward
 * sums  sum(dz), sum(dz * xhat)  over the stored rows (dz = dy masked by the recomputed ReLU), and the backward apply
 * dx = gamma * rstd * (dz - dbeta/count - xhat * dgamma/count) with caller-provided totals. */
int tmae_bn_stats(const void* x, int dtype, int64_t m, int c, double count, float eps, float* mean, float* var,
                  float* rstd, void* ws, size_t ws_bytes, void* stream);
/* y = relu?((x - mean) * rstd * gamma + beta) with caller-provided statistics (the apply half of tmae_bn_relu_fwd): for
 * statistics merged over ranks first -- torch.nn.SyncBatchNorm, tools/train.py:74,244-245 (--sync_bn). */
int tmae_bn_apply(const void* x, int dtype, int64_t m, int c, const float* mean, const float* rstd, const float* gamma,
                  const float* beta, int relu, void* y, void* stream);
int tmae_bn_bwd_sums(const void* dy, const void* x, int dtype, int64_t m, int c, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int relu, float* sum_dz, float* sum_dz_xhat, void* ws,
                     size_t ws_bytes, void* stream);
/* The same pass with a third total, sum_dy [c] = the column sums of dy before the ReLU mask (what tmae_deblock_bn_tail needs to
 * price the inactive cells of a dense map; it was a separate tmae_column_sums pass over dy).  Workspace: 2 x tmae_bn_workspace(m, c). */
int tmae_bn_bwd_sums3(const void* dy, const void* x, int dtype, int64_t m, int c, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int relu, float* sum_dy, float* sum_dz, float* sum_dz_xhat,
                      void* ws, size_t ws_bytes, void* stream);
int tmae_bn_bwd_apply(const void* dy, const void* x, int dtype, int64_t m, int c, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int relu, const float* dbeta,
                      const float* dgamma, double count, void* dx, void* stream);

/* Decoder head, SiamWCA_MAE.dense_conv (SiamWCA_MAE.py:231-253; modules :79-98): SparseConvTensor.dense() ->
 * ConvTranspose2d(k = s, stride s, no bias) -> BatchNorm2d -> ReLU -> torch.cat(dim=1), without materialising the
 * dense intermediates.  v [m, s*s*cout] = feat @ W (column order (dy, dx, cout)) holds the deconv output on the
 * active input cells of the [batch, ys, xs] grid; every other output cell is exactly zero before the norm.
 *   tmae_deblock_scatter: out[b, y, x, coff:coff+cout] (channels-last, row pitch ldc elements) =
 *       relu(v[grid[b, y/s, x/s], (y%s, x%s), :] * rstd*gamma + beta - mean*rstd*gamma), v := 0 where grid < 0.
 *   tmae_deblock_gather:  g [m, s*s*cout] = rows of the dense gradient dcat at the active cells (indices [m,3] b,y,x).
 *   tmae_column_sums:     out[c] = sum over rows of x [rows, c] (c % 64 == 0, c <= 512), fixed-order. */
int tmae_deblock_scatter(const void* v, int dtype, const int32_t* grid, int batch, int ys, int xs, int s, int cout,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, void* out,
                         int ldc, int coff, void* stream);
/* All sources of the concat buffer in ONE launch (whole ldc-wide rows per store burst instead of one channel slice per launch):
 * HOST arrays of n_src <= 4 entries -- v[i], grid[i], ys[i], xs[i], s[i], cout[i], mean[i], rstd[i], gamma[i], beta[i] as in
 * tmae_deblock_scatter; the slices lie side by side in source order (sum of cout = ldc), every source covers the same dense
 * grid (ys[i] s[i] = Y, xs[i] s[i] = X); strides are powers of two. */
int tmae_deblock_scatter_multi(int n_src, const void* const* v, int dtype, const int32_t* const* grid, int batch, const int* ys,
                               const int* xs, const int* s, const int* cout, const float* const* mean, const float* const* rstd,
                               const float* const* gamma, const float* const* beta, void* out, int ldc, void* stream);
int tmae_deblock_gather(const void* dcat, int dtype, int ldc, int coff, const int32_t* indices, int64_t m, int ys,
                        int xs, int s, int cout, void* g, void* stream);
size_t tmae_column_sums_workspace(int64_t rows, int c);
int tmae_column_sums(const void* x, int dtype, int64_t rows, int c, float* out, void* ws, size_t ws_bytes,
                     void* stream);
/* Backward of that norm, the inactive output cells' share (they all hold z0 = beta - mean rstd gamma, xhat0 = -mean rstd):
 *   rest = (s_all - s_act) [z0 > 0],  dbeta = sum_dz + rest,  dgamma = sum_dzx + rest xhat0     (all [c] f32)
 * s_all / s_act: column sums of dy over all cells / over the active cells (tmae_column_sums), sum_dz / sum_dzx: the active
 * cells' sums from tmae_bn_bwd_sums.  torch autograd of BatchNorm2d + ReLU on the dense map in the reference. */
int tmae_deblock_bn_tail(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* s_all,
                         const float* s_act, const float* sum_dz, const float* sum_dzx, int c, float* dbeta, float* dgamma,
                         void* stream);
/* The whole BatchNorm backward of one source of the fused decoder head (the backward of tmae_deblock_scatter up to the
 * deconvolution's rows), reading the source's gradient rows IN PLACE from the gradient of the concat buffer: v [m s^2, cout] the
 * deconvolution's rows at the m active voxels indices [m, 3], dcat [batch, ys s, xs s, ldc] (this source at channel offset coff),
 * s_all [cout] the column sums of dcat over all cells for these channels, count = batch * ys s * xs s.  Outputs dv [m s^2, cout]
 * (gradient of v), dgamma, dbeta.  s in {1, 2, 4}, cout in {64, 128, 256}.  Replaces tmae_deblock_gather + tmae_column_sums +
 * tmae_bn_bwd_sums + tmae_deblock_bn_tail + tmae_bn_bwd_apply.  ws: tmae_deblock_bn_bwd_workspace(m, s, cout). */
size_t tmae_deblock_bn_bwd_workspace(int64_t m, int s, int cout);
int tmae_deblock_bn_bwd(const void* dcat, int dtype, int64_t ldc, int coff, const int32_t* indices, int64_t m, int ys, int xs, int s,
                        int cout, const void* v, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        const float* s_all, double count, void* dv, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                        void* stream);

/* bf16 copies of fp32 parameter matrices after the optimizer step (what autocast reads: torch casts a weight on every
 * use; the dX GEMMs of the token-list Linears additionally want W^T): ONE launch for a list of matrices.
 * table (device, 8-byte aligned): `count` entries of five int64 -- src (fp32 [n,k]), dst (bf16 [n,k]), dstT (bf16 [k,n] or
 * 0), n | k << 32, index of the entry's first 32 x 32 tile -- entries ordered by that index; total_tiles = their sum. */
int tmae_multi_cast_transpose(const void* table, int count, int64_t total_tiles, void* stream);

/* The optimizer step of the recipe for ALL parameter tensors in one launch: decoupled weight decay p *= 1 - wd*lr for every
 * entry (OptimWrapper.step, tools/train_utils/optimization/fastai_optim.py:139-150: true_wd, bn_wd), then torch.optim.Adam's
 * update (fastai_optim.py:151 -> Adam.step; amsgrad / maximize off, weight_decay 0 inside Adam) for the entries that
 * have a gradient.  table (device, 8-byte aligned): entries of EIGHT int64 -- p, grad (0: decay only), exp_avg, exp_avg_sq,
 * the per-parameter step tensor (1 float on the device, receives the entry's step; 0: none), numel | first chunk << 40, a
 * bf16 copy of the parameter to refresh (what the autocast forward reads; 0: none), the entry's step number after the
 * increment (>= 1 for entries with a gradient; Adam keeps the step per parameter, so tensors that missed a gradient once
 * lag behind) -- where a chunk = 4096 elements and the entries are ordered by their first chunk; chunk_tensor (device
 * int32 [total_chunks]) = the entry of every chunk.  The bias corrections are taken in double per entry, as torch takes
 * them on the host. */
int tmae_adam_step(const void* table, const int32_t* chunk_tensor, int64_t total_chunks, float lr, float beta1, float beta2,
                   float eps, float weight_decay, void* stream);

/* Running statistics of all BatchNorm layers of a forward pass in one launch (torch.nn.BatchNorm's training-mode update:
 * running = (1 - momentum) * running + momentum * batch statistic, unbiased variance; num_batches_tracked += 1 -- the norms
 * of network_utils.py:31, spconv_utils.py:50-54, SiamWCA_MAE.py:91-115).  bufs (device): nbufs entries of four int64 --
 * running buffer (fp32), numel, index of its first update, number of its updates (applied in order: a module used twice);
 * updates (device): entries of two int64 -- batch statistic (fp32 [numel]), float bits keep | scale << 32 (running =
 * running * keep + statistic * scale); counters (device): ncounters pointers to int64 num_batches_tracked, each += 1. */
int tmae_bn_running_update(const void* bufs, int nbufs, const void* updates, const void* counters, int ncounters,
                           void* stream);

/* Token-list Linear in bf16 (fp32 accumulate):  y[m,n] = x[m,k] . w[n,k]^T (+ bias[n]) -- the in-/out-projections
 * and FFN layers of EncoderLayer (sst_basic_block.py:45-83, F.linear) and, on w^T, their input gradients.
 * k in {32, 64, 128, 256, 512} (32, 64: the VFE MLP, temporal_dyn_vfe.py:110-112), n a multiple of 64; ldx / ldy = row pitches in elements (column slices of packed buffers are
 * fine); all pointers 16-byte aligned (y, bias: 8); bias [n] bf16 is required (zeros for none); y must span
 * < 2^31 bytes.  x is read once, y written once. */
int tmae_token_gemm(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                    int64_t ldy, void* stream);
/* In-place form  y[m,n] += x[m,k] . w[n,k]^T + bias[n]  -- torch.Tensor.addmm_ of the Linear input gradients whose input has
 * a second consumer (ops.proj_fork: the FFN's first Linear, sst_basic_block.py:81-83, autograd's AccumulateGrad add in
 * the reference) and of the attention in-projections (sst_basic_block.py:45-47 backwards).  Shapes: (k, n) = (512, 256),
 * (256, 128), (768, 256), (384, 128), (256, 256) or (128, 128), m >= 32768; anything else returns TMAE_EARG and the caller
 * keeps the library's addmm_.  Same pointer rules as tmae_token_gemm; y is read and written once. */
int tmae_token_gemm_acc(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                        int64_t ldy, void* stream);
/* Residual form  y[m,n] = res[m,n] + x[m,k] . w[n,k]^T + bias[n]  (res: the pitch of y, 16-byte aligned; res != y) -- `src +
 * linear2(act)` of EncoderLayer.forward (sst_basic_block.py:81-83): the residual add in front of norm2 rides on the second Linear
 * of the FFN (one rounding of the sum instead of two, and the norm reads one tensor instead of two).  (k, n) = (512, 256) or
 * (256, 128), m >= 32768; TMAE_EARG otherwise. */
int tmae_token_gemm_res(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, const void* res,
                        void* y, int64_t ldy, void* stream);

/* The FFN's first Linear with its activation (sst_basic_block.py:81: activation(linear1(src)), exact erf GELU):
 * y = x . w^T + bias (the pre-activation, which the backward needs) AND y_gelu = gelu(y), same shape and pitch, written
 * by the same launch -- the separate elementwise GELU pass (read and write of [m, n]) becomes one extra write.
 * (k, n) = (256, 512) or (128, 256), m >= 32768; anything else returns TMAE_EARG and the caller runs tmae_token_gemm plus
 * its own activation pass.  Same pointer rules as tmae_token_gemm (y_gelu: 16-byte aligned). */
int tmae_token_gemm_gelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                         void* y_gelu, int64_t ldy, void* stream);
/* Same GEMM with the backward of the exact (erf) GELU fused into the epilogue: y = (x . w^T + bias) * gelu'(aux),
 * aux [m,n] bf16 with y's pitch = the pre-activation saved by the forward (the FFN of EncoderLayer,
 * sst_basic_block.py:81: linear2(gelu(linear1(src))): dX of linear2 and GeluBackward in one pass). */
int tmae_token_gemm_dgelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias,
                          const void* aux, void* y, int64_t ldy, void* stream);
/* The attention in-projections with the position embedding folded into the GEMM (q = k = src + pos, v = src:
 * WindowAttention.forward sst_basic_block.py:45-52, wca_block.py:50-60; embedding of SSTInputLayer.get_pos_embed,
 * spt_backbone.py:186-224).  The embedding depends only on the token's cell (xc, yc) inside its 8 x 8 window and is
 * separable, pos[cell] = [ex[xc] | ey[yc]], so (x + pos) W^T = x W^T + Tx[xc] + Ty[yc]:
 *   y [m,n] = [x | onehot(cells)] . w_aug[n, k+32]^T + bias,
 * w_aug = [W | Tx_hi Ty_hi Tx_lo Ty_lo] (bf16; T* [n,8], hi + lo = the fp32 table, zero rows for projections that take
 * no position, e.g. v), cells [m] u8 = xc | yc << 3 (tmae_window_cells).  k in {128, 256}, n % 64 == 0; otherwise as
 * tmae_token_gemm.  One streaming pass over x; no [m,d] x + pos tensor exists. */
int tmae_token_gemm_pos(const void* x, int64_t ldx, int64_t m, int k, const void* w_aug, int n, const void* bias,
                        const uint8_t* cells, void* y, int64_t ldy, void* stream);
/* cells [cells_len >= m] u8 (entries m .. cells_len-1 are written as zeros: the padding tmae_linear_wgrad_cells may read)
 * and onehot [m,16] bf16 (columns 0..7 one-hot xc, 8..15 one-hot yc) of the tokens `indices` [m,3] (b,y,x)
 * for window shape (wy, wx) <= 8 and the shift of the layer; onehot (may be NULL) is the explicit operand of the
 * position part of the in-projection weight gradient, dY^T . onehot -- tmae_linear_wgrad_cells computes that product
 * from `cells` inside the weight-gradient pass. */
int tmae_window_cells(const int32_t* indices, int64_t m, int64_t cells_len, int wy, int wx, int do_shift, uint8_t* cells,
                      void* onehot, void* stream);

/* ---- fine-tune path (BASELINE configs[4]): CenterHead ------------------------------------------------------
 * Target assignment of one head (CenterHead.assign_targets / assign_target_of_single_head, center_head.py:107-231;
 * centernet_utils.gaussian_radius / draw_gaussian_to_heatmap, centernet_utils.py:9-74).  gt_boxes [batch, nbox, ncode]
 * f32 rows (x, y, z, dx, dy, dz, heading, ..., class) with class 1..num_class_all (0 = padding row); cls_map
 * [num_class_all + 1] i32 maps the global class id to the index inside this head or -1.  Outputs (ZEROED by the caller):
 * heatmap [batch, num_class_head, H, W] f32, target_boxes [batch, nmax, ncode] f32, inds / mask [batch, nmax] i64.
 * Slot k of a sample = rank of the box among the sample's boxes of this head (the reference's loop index). */
int tmae_centerhead_targets(const float* gt_boxes, int batch, int nbox, int ncode, const int32_t* cls_map,
                            int num_class_all, int num_class_head, int H, int W, float pcr_x, float pcr_y, float vs_x,
                            float vs_y, float stride, int nmax, double overlap, int min_radius, float* heatmap,
                            float* target_boxes, int64_t* inds, int64_t* mask, void* stream);
/* CenterNet focal loss on logits (loss_utils.neg_loss_cornernet, loss_utils.py:273-309, applied to
 * clamp(sigmoid(x), 1e-4, 1-1e-4), center_head.py:233-244).  out4 = {loss, num_pos, pos_sum, neg_sum} f32;
 * backward: dlogits = grad_out * d loss / d logits (stats4 = out4 of the forward). */
size_t tmae_focal_loss_workspace(int64_t n);
int tmae_focal_loss_fwd(const void* logits, int dtype, const float* target, int64_t n, float* out4, void* ws,
                        size_t ws_bytes, void* stream);
int tmae_focal_loss_bwd(const void* logits, int dtype, const float* target, int64_t n, const float* stats4,
                        const float* grad_out, void* dlogits, void* stream);

/* Rotated boxes (x, y, z, dx, dy, dz, heading), pcdet/ops/iou3d_nms (iou3d_nms_utils.py:31-99, iou3d_nms_kernel.cu).
 * tmae_boxes_pairwise: out [na, nb] f32; mode 0 = BEV overlap area (boxes_overlap_bev_gpu), 1 = BEV IoU
 * (boxes_iou_bev_gpu), 2 = 3-D IoU (boxes_iou3d_gpu); mode | 4: PAIRED, out [na] = f(a[i], b[i]) with na == nb (the
 * diagonal that IoULossCenterNet takes of the full matrix, loss_utils.py:411-420).
 * tmae_nms_bev: nms_gpu on boxes ALREADY sorted by descending score: keep [<= n] i64 = indices (into the sorted
 * list) of the boxes that survive, in score order; num_keep [1] i32.  Greedy pass runs on the device. */
int tmae_boxes_pairwise(const float* boxes_a, int na, const float* boxes_b, int nb, int mode, float* out, void* stream);
size_t tmae_nms_bev_workspace(int n);
int tmae_nms_bev(const float* boxes_sorted, int n, float thresh, int64_t* keep, int32_t* num_keep, void* ws,
                 size_t ws_bytes, void* stream);

/* ---- data path in front of the step (ONCETemporalDataset.__getitem__ / prepare_data, once_temporal_dataset.py:139-330)
 * One frame of one sample: drop ego-vehicle returns (|x| and |y| < ego_radius), apply the previous->current alignment
 * (r1t1 = [R1 row-major 9 | t1 3] float64 or NULL; m2 = first three rows of the inverse current pose, 12 float64, or
 * NULL -- once_utils.convert_prv_frame_to_cur), world flip / rotation (cos, sin of the fp32 angle) / scaling
 * (data_augmentor.py:55-142), keep x, y inside [min, max] (common_utils.mask_points_by_range), write the kept rows in
 * input order as (batch_idx, x, y, z, features...) into out [<= n, row + 1]; count [1] i32 = kept rows. */
size_t tmae_frame_prepare_workspace(int64_t n);
int tmae_frame_prepare(const float* points, int row, int64_t n, const double* r1t1, const double* m2, float ego_radius,
                       int flip_x, int flip_y, float cosa, float sina, float scale, float xmin, float ymin, float xmax,
                       float ymax, int batch_idx, float* out, int32_t* count, void* ws, size_t ws_bytes, void* stream);
/* The same with gt_sampling's point removal (database_sampler.py:201-205: remove_points_in_boxes3d with REMOVE_EXTRA_WIDTH ->
 * roiaware_pool3d.cpp:119-140): points that lie, AFTER the alignment and BEFORE the augmentation, inside one of n_boxes <= 64
 * boxes are dropped.  remove_boxes (device, [n_boxes, 8] float64, prepared by the host): cx, cy, cz, (float)cos(-heading),
 * (float)sin(-heading), dz / 2, dx / 2 + 1e-2f, dy / 2 + 1e-2f -- the reference's fp32 rotation and double thresholds. */
int tmae_frame_prepare_boxes(const float* points, int row, int64_t n, const double* r1t1, const double* m2, float ego_radius,
                             int flip_x, int flip_y, float cosa, float sina, float scale, float xmin, float ymin, float xmax,
                             float ymax, int batch_idx, const double* remove_boxes, int n_boxes, float* out, int32_t* count,
                             void* ws, size_t ws_bytes, void* stream);

/* ---- box calibration probes (bench.py `peak_measured`; SURVEY.md section 7: measured achievable peaks on the box next
 * to the spec peaks -- no reference counterpart).  tmae_probe_copy: a 16-bytes-per-lane streaming copy of `bytes` (a
 * multiple of 16) from src to dst (nontemporal != 0: nt loads and stores): HBM bytes moved = 2 x bytes.  tmae_probe_mfma: `iters` x 16 back-to-back
 * v_mfma_f32_16x16x32_bf16 per wave on random register operands, one 512-thread workgroup per CU; *flops (host, optional)
 * receives the FLOPs of the launch; sink = 1 float on the device (never written in practice). */
int tmae_probe_copy(const void* src, void* dst, int64_t bytes, int nontemporal, void* stream);
int tmae_probe_mfma(int iters, float* sink, int64_t* flops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TMAE_HIP_H */
