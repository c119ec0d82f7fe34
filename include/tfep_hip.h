/*
 * tfep_hip.h -- C ABI of libtfep_hip.so: the MI355X (gfx950) kernels of the
 * tfep.nn normalizing-flow hot path.
 *
 * The reference (andrrizzi/tfep @ 2024-12-20) is pure Python on PyTorch and has
 * no FFI of its own; each entry point below replaces one reference function of
 * the path (cited as file:line relative to the reference root) and is what a
 * ctypes / cffi binding added to that function would call.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 / int32 unless marked "host";
 *   - tensors are dense row-major; `ld*` arguments are row strides in elements;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call only enqueues work on that stream and never synchronises;
 *   - every function returns 0 on success, a negative tfep_status otherwise, and
 *     never throws; tfep_last_error() returns a host string for the calling thread;
 *   - "parameters" follow the reference's conditioner-output layout unless a
 *     tfep_param_layout says otherwise: column p*D + f of a (B, P*D) matrix is
 *     parameter p of feature f (transformers/spline.py:351-352, affine.py:138-141);
 *   - log_det_J outputs are (B,) float32; with accumulate != 0 the kernel ADDS to
 *     the existing values (the `cumulative_log_det_J += log_det_J` of
 *     flows/sequential.py:66), otherwise it overwrites them.
 */
#ifndef TFEP_HIP_H
#define TFEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFEP_HIP_ABI_VERSION 8

typedef enum tfep_status {
    TFEP_OK = 0,
    TFEP_ERR_INVALID_ARGUMENT = -1,
    TFEP_ERR_UNSUPPORTED = -2,
    TFEP_ERR_LAUNCH = -3
} tfep_status;

/* ABI version of the loaded library (== TFEP_HIP_ABI_VERSION it was built with). */
int tfep_hip_abi_version(void);
/* Host string describing the last error on the calling thread ("" if none). */
const char* tfep_last_error(void);

/* ------------------------------------------------------------------------- */
/* Layout of transformer parameters in memory.                                */
/*   parameter p of feature f of sample b lives at                            */
/*       params[b * ld + p * stride_p + f * stride_f]                         */
/*   reference layout: ld = P*D, stride_p = D, stride_f = 1.                  */
/* ------------------------------------------------------------------------- */
typedef struct tfep_param_layout {
    int64_t ld;
    int64_t stride_p;
    int64_t stride_f;
} tfep_param_layout;

/* ------------------------------------------------------------------------- */
/* Masked linear layers (tfep/nn/masked.py)                                   */
/* ------------------------------------------------------------------------- */

/*
 * Effective masked weight, recomputed from its parametrisation.
 *   weight_g != NULL :  W[o,i] = mask[o,i] != 0 ? v[o,i] * (g[o] / ||v[o,:]||_2) : 0
 *                       replaces MaskedWeightNorm.compute_weight + _ApplyMask
 *                       (masked.py:369-371, :433-439; NaN-safe for fully-masked rows)
 *   weight_g == NULL :  W[o,i] = v[o,i] * mask[o,i]      (masked.py:270; mask may be NULL)
 * The result is written permuted and zero-padded for the GEMM kernels:
 *   w_out[row_of_out[o] * ldw + col_of_in[i]] = W[o,i]
 * row_of_out / col_of_in may be NULL (identity).  clear != 0: padding rows/cols of w_out (n_rows_padded x ldw) that no
 * (o,i) maps to are set to 0 first (a kernel, not hipMemsetAsync); clear == 0: the caller zeroed the buffer once and only
 * ever packs the same layer into it (every mapped entry is rewritten, the padding is never touched).
 * col_cut (or NULL): mask rows that are prefixes of the packed columns, mask[o][i] == (col_of_in[i] < col_cut[o]), as in
 * tfep_masked_weight_prepare_split: the mask is not read.
 */
int tfep_masked_weight_prepare(const float* weight_v, const float* weight_g, const float* mask,
                               int out_features, int in_features,
                               const int32_t* row_of_out, const int32_t* col_of_in, const int32_t* col_cut, int clear,
                               float* w_out, int n_rows_padded, int64_t ldw, void* stream);
/* The same packing for a layer whose mask rows are prefixes of its packed columns (col_cut, see
 * tfep_masked_weight_prepare_split): the mask is not read, a row of weight_v goes through LDS once (in_features * 4 <=
 * 64 KiB), and only the live prefix of each packed row is written, in whole 32-byte groups -- w_out must have been
 * zeroed when it was allocated and always hold this layer (ldw >= in_features rounded up to 8, 16-byte aligned rows).
 * in_of_col: packed column -> input feature (the inverse of col_of_in), or NULL.  Bit-identical to
 * tfep_masked_weight_prepare on the entries it writes.  (masked.py:369-371, :433-439) */
int tfep_masked_weight_prepare_prefix(const float* weight_v, const float* weight_g, int out_features, int in_features,
                                      const int32_t* row_of_out, const int32_t* in_of_col, const int32_t* col_cut,
                                      float* w_out, int n_rows_padded, int64_t ldw, void* stream);

/*
 * Per column-tile bounding range of the mask non-zeros, in PACKED coordinates:
 *   k_ranges[2*t], k_ranges[2*t+1] = [k_begin, k_end) over packed rows t*tile_n .. (t+1)*tile_n - 1,
 * rounded outwards to multiples of tile_k (k_end clipped to k_padded); an all-zero tile gets [0, 0).
 * The GEMM skips everything outside the range: the autoregressive masks of MADE are block
 * triangular once the hidden units are sorted by degree (conditioners/made.py:286-329,
 * masked.py:90-99).  The ranges depend on the mask buffer only -- compute once, reuse.
 * row_of_out / col_of_in: the same permutations given to tfep_masked_weight_prepare (or NULL).
 */
int tfep_mask_k_ranges(const float* mask, int out_features, int in_features,
                       const int32_t* row_of_out, const int32_t* col_of_in,
                       int tile_n, int tile_k, int n_tiles, int k_padded,
                       int32_t* k_ranges, void* stream);

/*
 * y = act(x W^T + b)      replaces MaskedLinearFunc.forward + F.linear (masked.py:265-277)
 *                         and the ELU of MADE (conditioners/made.py:320).
 *   x (B, >= k_padded) row stride ldx, columns [K, k_padded) must be ZERO (k_padded = K rounded
 *   up to tfep_masked_linear_tile_k());  w (n_rows_w, >= k_padded) row stride ldw, packed by
 *   tfep_masked_weight_prepare;  bias (N) in packed row order or NULL;  y (B, .) row stride ldy.
 *   k_ranges: output of tfep_mask_k_ranges with tile_n = tfep_masked_linear_tile_n(),
 *   tile_k = tfep_masked_linear_tile_k(), or NULL for the full K range.
 *   tile_order: optional (n_tiles) int32 permutation of the column tiles: workgroups are launched in
 *   this order (pass the tiles sorted by descending k-range so the longest run first); NULL = 0..n-1.
 *   col_map: optional (N) int32; packed output column j is stored at y[:, col_map[j]], skipped if
 *   col_map[j] < 0; NULL = identity.   act: 0 = identity, 1 = ELU(alpha = 1).
 *   tile_n: column-tile width, 0 / tfep_masked_linear_tile_n() (default) or
 *   tfep_masked_linear_narrow_tile_n() for few output rows (the per-degree row slices of the blocked
 *   autoregressive inverse, flows/autoregressive.py:179-229); k_ranges must use the same tile_n.
 *   Both operands must be 16-byte aligned with row strides that are multiples of 4 floats.
 * fp32 MFMA (v_mfma_f32_16x16x4_f32): exact fp32 products, fp32 accumulation.
 */
int tfep_masked_linear_forward(const float* x, int64_t ldx, const float* w, int64_t ldw,
                               const float* bias, const int32_t* k_ranges, const int32_t* tile_order,
                               const int32_t* col_map, float* y, int64_t ldy, int B, int N, int n_rows_w,
                               int k_padded, int act, int tile_n, void* stream);
int tfep_masked_linear_tile_n(void);
int tfep_masked_linear_narrow_tile_n(void);
/* Widest column tile of the split-f16 kernel (400: the fused spline kernel's tile for a plain linear product, tile_n of
 * tfep_gemm_desc with split = 1, act = 0; k_ranges / tile_live / tile_list then count tiles of this width). */
int tfep_split_wide_tile_n(void);
/* ... and a 208-column one (same conditions): two of them cover the 400 output rows of a block of 16 degrees of an 8-bin
 * spline layer -- the output-layer GEMM of the blocked inverse (flows/autoregressive.py:179-229) -- with 4 % padding. */
int tfep_split_half_wide_tile_n(void);

/*
 * General form of the same GEMM, used by the backward pass (MaskedLinearFunc.backward,
 * masked.py:279-302: grad_input = g W_masked, grad_weight = (g^T x) o mask):
 *   y (+)= act(x w^T + bias) [* elu'(elu_grad_of)]
 *   elu_grad_of: optional saved ELU OUTPUT h with the layout of y; the result is multiplied by
 *     elu'(z) = h > 0 ? 1 : h + 1   (ELU backward of conditioners/made.py:320 fused into grad_input);
 *   accumulate: add into y (gradient accumulation over batch chunks);
 *   tile_live: optional (ceil(B/tile_m) x ceil(N/tile_n)) bytes, 0 = output tile entirely masked, skipped
 *     (grad_weight of a block-triangular mask).
 *   tile_list / n_tile_list: optional launch order of the output tiles, position p -> (row tile, column tile) =
 *     (tile_list[2p], tile_list[2p + 1]), negative = none; tiles not listed are not computed.  Workgroups go
 *     round-robin to the 8 XCDs: list the live tiles of a block-triangular grad_weight in groups of 32 neighbours,
 *     group k at positions ((k / 8) * 32 + w) * 8 + k % 8, and every XCD gets the same share (wide tile only,
 *     no tile_order / k_split).
 *   pre_add: partial pre-activations added before the activation (two-level blocked inverse: the block's
 *     contribution of all earlier degrees is computed once, each degree adds only its own block's part).
 * Other fields as in tfep_masked_linear_forward.
 */
typedef struct tfep_gemm_desc {
    const float* x; int64_t ldx;
    const float* w; int64_t ldw;
    const float* bias;
    const int32_t* k_ranges;
    const int32_t* tile_order;
    const int32_t* col_map;
    float* y; int64_t ldy;
    int32_t B, N, n_rows_w, k_padded, act, accumulate;
    const float* elu_grad_of; int64_t ld_elu_grad_of;
    const uint8_t* tile_live;
    const float* pre_add; int64_t ld_pre_add;   /* optional, layout of y: y = act(x w^T + bias + pre_add) */
    int32_t tile_n;                             /* 0 / wide (default), tfep_masked_linear_narrow_tile_n(), or (split) tfep_split_wide_tile_n() */
    int32_t split;                              /* 1: x and w are split-f16 rows (tfep_split_rows), wide tile only */
    const float* x_inv_scale;                   /*    (B) per-row 1/scale of x                                     */
    const float* w_inv_scale;                   /*    (1) 1/scale of w                                             */
    int32_t split_out;                          /* with split = 1, act = 1: y receives ELU(...) as SPLIT rows for the */
    float* y_inv_scale;                         /*   next layer (ldy = pitch in 4-byte elements); per-row 1/scale (B)  */
    const float* w_l1max;                       /*   scale from the bound max|x_row| * w_l1max + bias_absmax:          */
    const float* bias_absmax;                   /*   inv_scale[2] of tfep_masked_weight_prepare_split, max |bias|      */
    int32_t k_split;                            /* > 1: split-K for products with few output tiles: slice s of the  */
    int64_t slab_stride;                        /*   k-range writes its partial sums to y + s * slab_stride (bias and */
                                                /*   pre_add in slice 0); the caller adds the k_split slabs           */
    const int32_t* tile_list;                   /* optional (2 * n_tile_list): launch position -> (row tile, column tile) */
    int32_t n_tile_list;
} tfep_gemm_desc;
int tfep_masked_linear_gemm(const tfep_gemm_desc* desc, void* stream);
int tfep_masked_linear_tile_k(void);
int tfep_masked_linear_tile_m(void);

/* ------------------------------------------------------------------------- */
/* Transformers (tfep/nn/transformers)                                        */
/* ------------------------------------------------------------------------- */

/* y = x * exp(a) + b, ldj = sum_f a     (affine.py:321-323; inverse :361-363)
 * parameter 0 = shift b, parameter 1 = log-scale a (affine.py:136-141). */
int tfep_affine_forward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout,
                        float* y, int64_t ldy, float* log_det_J, int accumulate,
                        int B, int D, void* stream);
int tfep_affine_inverse(const float* y, int64_t ldy, const float* params, tfep_param_layout layout,
                        float* x, int64_t ldx, float* log_det_J, int accumulate,
                        int B, int D, void* stream);

/* y = x + b with optional periodic wrap, ldj = 0  (affine.py:366-456).
 * periodic_mask: (D) int32, non-zero for wrapped features, or NULL.
 * sign = +1 forward, -1 inverse. */
int tfep_volume_preserving_shift(const float* x, int64_t ldx, const float* shift, int64_t ldp,
                                 const int32_t* periodic_mask, float lower, float upper, int sign,
                                 float* y, int64_t ldy, int B, int D, void* stream);

/* Rational-quadratic neural spline (transformers/spline.py:29-650). */
typedef struct tfep_spline_desc {
    const float* x0;      /* (D) first knot of the input domain   (spline.py:153)  */
    const float* xf;      /* (D) last knot of the input domain    (spline.py:154)  */
    const float* y0;      /* (D) first knot of the output domain  (spline.py:156)  */
    const float* yf;      /* (D) last knot of the output domain   (spline.py:157)  */
    int32_t n_bins;       /* K                                                      */
    int32_t circular;                 /* spline.py:158, :236-238, :257-259          */
    int32_t identity_boundary_slopes; /* spline.py:159, :378-380                    */
    int32_t learn_lower_bound;        /* spline.py:160, :387-410                    */
    int32_t learn_upper_bound;        /* spline.py:161                              */
    float min_bin_size;               /* spline.py:162, :394-395                    */
    float min_slope;                  /* spline.py:163, :414-415                    */
} tfep_spline_desc;

/* Parameters per feature for a descriptor (spline.py:165-182). */
int tfep_spline_n_parameters_per_feature(const tfep_spline_desc* desc);

/* NeuralSplineTransformer.forward / .inverse (spline.py:184-261): parameter split and
 * normalisation (_get_parameters :319-417), circular shift, strict-'>' bin search
 * (_assign_bins :567-650), RQ evaluation (:478-494 / :521-536) and log|det J| (:546-564).
 * Outside the spline domain the map is linear along the boundary slope. */
int tfep_spline_forward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout,
                        const tfep_spline_desc* desc, float* y, int64_t ldy,
                        float* log_det_J, int accumulate, int B, int D, void* stream);
int tfep_spline_inverse(const float* y, int64_t ldy, const float* params, tfep_param_layout layout,
                        const tfep_spline_desc* desc, float* x, int64_t ldx,
                        float* log_det_J, int accumulate, int B, int D, void* stream);

/* Moebius transformer on `dimension`-vectors (moebius.py:374-478); inverse = forward with
 * -w (moebius.py:142-147): pass sign = -1.  D = n_vectors * dimension, dimension <= 8. */
int tfep_moebius_forward(const float* x, int64_t ldx, const float* params, int64_t ldp,
                         int dimension, float max_radius, int unit_sphere, int sign,
                         float* y, int64_t ldy, float* log_det_J, int accumulate,
                         int B, int D, void* stream);
/* The forward map of unit-sphere 2-vectors (dimension 2, unit_sphere 1, sign +1) that ALSO writes y as split-f16 rows for the
 * masked linear that reads it next (tfep_split_rows' format, row stride ld_split floats >= D, D a multiple of 8; columns past
 * D are not written: clear the buffer once) with the row scale of the bound |y| <= 1 (y_inv_scale[b] = 2^-14) -- the
 * conversion pass between two MAF layers (one read + one write of the batch) disappears.  Rows of x, params, y on 8-byte
 * boundaries. */
int tfep_moebius_forward_split_out(const float* x, int64_t ldx, const float* params, int64_t ldp, float max_radius,
                                   float* y, int64_t ldy, float* log_det_J, int accumulate,
                                   void* y_split, int64_t ld_split, float* y_inv_scale, int B, int D, void* stream);

/* PeriodicEmbedding.forward (embeddings/mafembed.py:112-145):
 * out = [x[:, nonperiodic]..., cos t0, sin t0, cos t1, sin t1, ...], t = (x - lower) * 2pi/(upper-lower). */
int tfep_periodic_embedding(const float* x, int64_t ldx, const int32_t* periodic_indices, int n_periodic,
                            const int32_t* nonperiodic_indices, int n_nonperiodic,
                            float lower, float upper, float* out, int64_t ldo, int B, void* stream);

/* Column gather / scatter used by AutoregressiveFlow / MixedTransformer for fixed and
 * conditioning features (flows/autoregressive.py:164-173, transformers/mixed.py:178-186):
 *   gather : dst[b, j] = src[b, idx[j]]      scatter: dst[b, idx[j]] = src[b, j]   (j < n_idx) */
int tfep_gather_columns(const float* src, int64_t lds, const int32_t* idx, int n_idx,
                        float* dst, int64_t ldd, int B, void* stream);
int tfep_scatter_columns(const float* src, int64_t lds, const int32_t* idx, int n_idx,
                         float* dst, int64_t ldd, int B, void* stream);

/* ------------------------------------------------------------------------- */
/* Fused MADE output layer + transformer (the headline kernel)                */
/* ------------------------------------------------------------------------- */

typedef enum tfep_fused_kind {
    TFEP_FUSED_AFFINE = 0,   /* P = 2                                         */
    TFEP_FUSED_SPLINE = 1    /* RQ spline of 8, 5 or 4 bins: see tfep_fused_supported */
} tfep_fused_kind;

/* Feature slots per 16-wide MFMA column group (16). */
int tfep_fused_tile_features(void);
/* 1 if (kind, desc) is supported by the fused kernel: the affine transformer; RQ splines of 8, 5 or 4 bins, plain or
 * circular, with or without identity boundary slopes and learnable bounds (11 .. 27 parameters per feature: every layout
 * of spline.py:165-182 at these bin numbers). */
int tfep_fused_supported(int kind, const tfep_spline_desc* desc);
/* Packed weight rows per column tile of the fused kernel (= tile_n for tfep_mask_k_ranges):
 * P * FT * 16 where FT feature groups of 16 slots share a tile (spline: P = parameters per feature, FT = 1;
 * affine: 2*8*16).
 * Packed row of (slot s, parameter p):  tile = s / (16*FT), ft = (s / 16) % FT, j = s % 16,
 *   row = tile * (P*FT*16) + (ft * P + p) * 16 + j. */
int tfep_fused_tile_columns(int kind, const tfep_spline_desc* desc);

/*
 * params = h W^T + b is formed tile by tile in MFMA accumulators and consumed in registers by
 * the transformer: the (B, P*D) parameter tensor never reaches HBM.
 *   h (B, >= k_padded) hidden activations (zero padded), w / bias_packed packed as above
 *   (already masked), k_ranges per column tile (tile_n = tfep_fused_tile_columns()), tile_order as in
 *   tfep_masked_linear_forward.
 *   x, y (B, .): transformer input / output; feature slot s reads x[:, feat_index[s]] and writes
 *   y[:, feat_index[s]] (feat_index[s] < 0: padding slot, nothing read or written);
 *   feat_tr[s] indexes desc->x0/xf/y0/yf (spline only).  n_feature_slots: multiple of 16*FT.
 *   ldj_partial: (n_feature_slots/16, B) float64 workspace; log_det_J (B,) is reduced from it.
 * Replaces MaskedLinear (last layer, masked.py:188-208) + NeuralSplineTransformer.forward
 * (spline.py:184-241) / AffineTransformer.forward (affine.py:51-77).
 */
int tfep_fused_output_transformer_forward(const float* h, int64_t ldh, const float* w, int64_t ldw,
                                          const float* bias_packed, const int32_t* k_ranges,
                                          const int32_t* tile_order,
                                          int kind, const tfep_spline_desc* desc,
                                          const float* x, int64_t ldx, float* y, int64_t ldy,
                                          const int32_t* feat_index, const int32_t* feat_tr,
                                          int n_feature_slots, double* ldj_partial,
                                          float* log_det_J, int accumulate,
                                          int B, int n_rows_w, int k_padded, void* stream);

/*
 * Split-precision operands for the MADE GEMMs (same results as the fp32 path to fp32 rounding, at the fp16
 * matrix-core rate).  A "split row" keeps the pitch of the fp32 row (4 bytes per element) and stores, per group of
 * 8 consecutive columns, 8 fp16 high halves then 8 fp16 low halves of  v * scale = hi + lo,  scale a power of two
 * with max |v * scale| in [2^14, 2^15):  per row (per_tensor = 0; inv_scale has `rows` entries) or one for the
 * whole matrix (per_tensor = 1; inv_scale has 2 entries: [0] = 1/scale, [1] = scratch).
 *   src (rows, >= cols) fp32;  dst (rows, ld_dst elements of 4 bytes), columns [cols, cols_padded) zero filled;
 *   cols_padded, ld_dst: multiples of tfep_split_tile_k() (32).
 * Activations use per-row scales, packed weights (tfep_masked_weight_prepare output) one scale.
 */
int tfep_split_tile_k(void);
int tfep_split_rows(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int64_t ld_dst,
                    int64_t cols_padded, float* inv_scale, int per_tensor, void* stream);

/* |.| reductions used for row-scale bounds (stream-ordered kernels only: no memset, no workspace -- safe to capture in a
 * HIP graph).  mode 0: out[row] = max_k |src[row, k]| (rows floats);  mode 1: out[0] = max_row sum_k |src[row, k]|. */
int tfep_abs_reduce(const float* src, int64_t ld_src, int64_t rows, int64_t cols, int mode, float* out, void* stream);

/* Data-aware guard of the split-f16 GEMMs (they stand in for the fp32 products of reference tfep/nn/masked.py:265-277,
 * 279-302): count[0] += number of rows of src (rows, >= cols) that hold a non-zero element below 2^-bits of the row's
 * largest magnitude, or a non-finite one.  With one power-of-two scale per row such elements are carried with fewer than 22
 * significant bits (bits = 19 for the fp16 hi + lo format); the host routes a flagged batch to the exact-fp32 kernels.
 * count is NOT cleared (several tensors may add to one counter); plain kernel, no workspace. */
int tfep_range_flag(const float* src, int64_t ld_src, int64_t rows, int64_t cols, int bits, int32_t* count, void* stream);
/* out[c] = max_r |src[r, c]| (cols floats; a NaN in a column survives): the per-feature magnitudes the guard above is run on
 * (one pass over the batch at HBM rate; plain kernels, no workspace). */
int tfep_column_absmax(const float* src, int64_t ld_src, int64_t rows, int64_t cols, float* out, void* stream);

/* Columns [col0, col0 + cols) (col0 % 8 == 0; whole groups of 8 are converted) of fp32 rows into the same columns of split
 * rows, with the per-row scale given by the caller (inv_scale[row], a power of two, e.g. from a bound on the row): for
 * operands that are filled incrementally -- the hidden-activation panels of the blocked inverse
 * (reference flows/autoregressive.py:179-229 recomputes every panel per degree instead). */
int tfep_split_columns_scaled(const float* src, int64_t ld_src, int64_t rows, int64_t col0, int64_t cols, void* dst,
                              int64_t ld_dst, const float* inv_scale, void* stream);

/* tfep_masked_weight_prepare writing split rows directly (one scale for the matrix, an upper bound of max |w|:
 * max |weight_g| with weight norm, max |weight_v| without).  in_of_col: PACKED column -> input column (the inverse
 * of tfep_masked_weight_prepare's col_of_in), or NULL for the identity.  Only the out_features real rows are
 * written (all k_padded columns of each): padding rows of w_split_out must already be zero.
 * inv_scale: 4 floats: [0] = 1/scale, [1] = scratch, [2] = max_j sum_k |w_jk| (row-L1 maximum of the effective
 * weights, used to bound the next layer's activations: tfep_gemm_desc.split_out), [3] unused.
 * col_cut (or NULL): when every mask row is a PREFIX in packed column order -- packed columns are sorted by the degree of
 * their input, so mask[o][in_of_col[c]] == (c < col_cut[o]) for the autoregressive masks of made.py:308-309 -- the mask
 * is not read at all (a third of the kernel's HBM traffic); the caller checks the prefix property once per mask.  With
 * col_cut only the live prefix [0, col_cut[o]) of each packed row (rounded up to 8 columns) is written: the rest of the
 * row must already be zero (a buffer zeroed once and only ever used for this layer and mask). */
int tfep_masked_weight_prepare_split(const float* weight_v, const float* weight_g, const float* mask, int out_features,
                                     int in_features, const int32_t* row_of_out, const int32_t* in_of_col, const int32_t* col_cut,
                                     void* w_split_out, int64_t ldw, int k_padded, float* inv_scale, void* stream);
/* The same pass writing BOTH forms of the effective weights: split rows (as above) and fp32 rows (w32_out, the packing of
 * tfep_masked_weight_prepare with the same row / column order: the same bits), from one read of weight_v.  Prefix masks
 * (col_cut) and rows of 8192 .. 16384 weights only (TFEP_ERR_UNSUPPORTED otherwise: use the two entry points).  The blocked
 * inverse (flows/autoregressive.py:179-229) needs the fp32 matrix for its block kernel and the split one for its GEMMs. */
int tfep_masked_weight_prepare_split_both(const float* weight_v, const float* weight_g, int out_features, int in_features,
                                          const int32_t* row_of_out, const int32_t* in_of_col, const int32_t* col_cut,
                                          void* w_split_out, int64_t ldw, float* w32_out, int64_t ldw32, int k_padded,
                                          float* inv_scale, void* stream);

/* tfep_fused_output_transformer_forward on split operands: h_split (B rows, per-row h_inv_scale) and w_split (one
 * w_inv_scale); every other argument as above.  k_ranges must be multiples of 32. */
int tfep_fused_output_transformer_forward_split(const void* h_split, int64_t ldh, const float* h_inv_scale,
                                                const void* w_split, int64_t ldw, const float* w_inv_scale,
                                                const float* bias_packed, const int32_t* k_ranges,
                                                const int32_t* tile_order, int kind, const tfep_spline_desc* desc,
                                                const float* x, int64_t ldx, float* y, int64_t ldy,
                                                const int32_t* feat_index, const int32_t* feat_tr, int n_feature_slots,
                                                double* ldj_partial, float* log_det_J, int accumulate, int B,
                                                int n_rows_w, int k_padded, void* stream);

/* The same launch for the forward of a TRAINING step (RQ splines): also writes the transformer parameters the backward
 * needs (MaskedLinear output incl. bias, masked.py:265-277) to theta_out (B, ld_theta), column slot * P + p for the live
 * feature slots, from the epilogue -- no separate output GEMM + spline kernel.  w_feature_major (must be 1): the rows of
 * w_split / bias_packed inside a column tile are feature-major (row = slot * P + p, the packing of the backward's
 * grad_input GEMM) instead of parameter-major (p * 16 + slot); k_ranges count the same 16 P-row tiles.  theta_out may be
 * NULL.  The live slots must be the first ones; padding slots (feat_index < 0) write nothing. */
int tfep_fused_saving_supported(const tfep_spline_desc* desc);      /* the plain and circular layouts of 8, 5 or 4 bins */
int tfep_fused_output_transformer_forward_split_saving(const void* h_split, int64_t ldh, const float* h_inv_scale,
                                                       const void* w_split, int64_t ldw, const float* w_inv_scale,
                                                       const float* bias_packed, const int32_t* k_ranges,
                                                       const int32_t* tile_order, const tfep_spline_desc* desc,
                                                       const float* x, int64_t ldx, float* y, int64_t ldy,
                                                       const int32_t* feat_index, const int32_t* feat_tr, int n_feature_slots,
                                                       double* ldj_partial, float* log_det_J, int accumulate, int B,
                                                       int n_rows_w, int k_padded, int w_feature_major, float* theta_out,
                                                       int64_t ld_theta, void* stream);

/* Diagnostic: matrix-pipe ceiling of this device for the split GEMM's instruction mix (cf. tfep_diag_mfma_peak). */
int tfep_diag_split_mfma_peak(float* scratch, int blocks, int iters, void* stream);
/* With TFEP_DIAG=16 the split kernels accumulate shader-cycle totals per phase; this reads and resets them
 * (synchronous): out[0] = k-loop, out[1] = epilogue, out[2] = workgroups counted, out[3] = sum of workgroup
 * lifetimes in 100 MHz real-time ticks. */
int tfep_diag_split_cycles(unsigned long long* out);

/*
 * Fused inner loop of the blocked autoregressive inverse (flows/autoregressive.py:179-229): one launch runs, for a
 * block of consecutive degrees, the whole per-degree chain -- hidden units of the degree in every layer (ELU),
 * the P parameters of the degree's features, the transformer inverse (affine.py:361-363 / spline.py:504-543) and
 * the update of x -- one thread per sample row.  The contribution of all EARLIER blocks to the block's rows must
 * already be in z[l] / zout (one tfep_masked_linear_gemm per layer, bias included).
 *   x (B, ldx), xpad (B, ldxpad): inverse output / zero-padded conditioner input, filled feature by feature;
 *   y (B, ldy): the transformed features; h[l] (B, ldh): hidden activations, units sorted by degree;
 *   w[l], wout: packed masked weights (tfep_masked_weight_prepare; wout rows grouped by degree: p * n_d + f);
 *   steps: n_steps records of tfep_inverse_block_step_ints() int32:
 *     per layer l < 4: [row0, n, kb, ke] -- units [row0, row0 + n) of layer l from the first ke features of the
 *     block (l = 0) or the packed columns [kb, ke) of layer l - 1 (l >= 1); then [out_row0, n_d, kb, ke, feat_off,
 *     member] (member: kind 3 only, else 0);
 *   feat_cols / feat_sel: column in x / index in y and in the spline tables of every feature of the block, in step order;
 *   conditioner inputs: a new feature enters xpad as itself, or as (cos, sin) of (x - emb_lower) * 2 pi / (emb_upper -
 *     emb_lower) under a PeriodicEmbedding (mafembed.py:112-145).  in_cols: xpad column of every input ENTRY of the
 *     block in step order (a periodic feature has two consecutive entries / columns); feat_in: first entry of each
 *     feature; feat_periodic: 1 / 0.  Layer-0 step records count entries, max_feats sizes the LDS for entries;
 *   cache_col0[l], cache_n_old[l]: first packed column of layer l held in LDS and how many of them earlier blocks
 *     computed (a multiple of 4; w[l], wout 16-byte aligned with ldw % 4 == 0: the kernel stages weight rows in LDS
 *     with 16-byte loads); cache_len / max_feats: LDS entries per layer / for the block's features
 *     (tfep_inverse_block_lds_bytes(n_layers, cache_len, max_feats) <= 160 KiB).
 *   kind: 0 affine, 1 spline (desc, n_bins <= 8), 2 Moebius (moebius.py:142-147: moebius_dim consecutive features of
 *     a degree form one vector, so every step's n_d must be a multiple of it), 3 RQ splines of several layouts (the
 *     members of a mixed transformer, mixed.py:64-68: `spline` is an array of n_spline_groups <= 8 descriptors, every
 *     step record names the member its features belong to -- a degree with features of m members is m steps, the
 *     later ones without hidden units -- and feat_sel indexes the member's own x0 / xf / y0 / yf as well as y, so
 *     the caller hands arrays laid out over all transformed features; a descriptor with n_bins = 0 is a plain shift
 *     member, affine.py:366-456 without periodic features: one parameter, x = y - parameter, log-det 0).
 *     log_det_J (B) is accumulated.
 */
typedef struct tfep_inverse_block_desc {
    int32_t B, n_layers, n_steps, kind;
    float* x; int64_t ldx;
    float* xpad; int64_t ldxpad;
    const float* y; int64_t ldy;
    float* h[4]; int64_t ldh[4];
    const float* z[4]; int64_t ldz[4];
    const float* zout; int64_t ldzout;
    int32_t z_slabs[4]; int64_t z_slab_stride[4];   /* z[l] / zout given as split-K slabs (tfep_gemm_desc.k_split): */
    int32_t zout_slabs; int64_t zout_slab_stride;   /*   the kernel adds slabs s = 0 .. n-1 at z + s * stride; 0/1 = plain */
    float* log_det_J;
    const float* w[4]; int64_t ldw[4];
    const float* wout; int64_t ldwout;
    const int32_t* steps;
    const int32_t* feat_cols;
    const int32_t* feat_sel;
    const int32_t* feat_in;
    const int32_t* feat_periodic;
    const int32_t* in_cols;
    float emb_lower, emb_upper;
    int32_t cache_col0[4], cache_n_old[4];
    int32_t cache_len, max_feats;
    const tfep_spline_desc* spline;
    int32_t moebius_dim, moebius_unit_sphere;
    float moebius_max_radius;
    int32_t rows_per_wave;      /* 64 (or 0): one sample row per lane; 16: four lanes per row, 4x the waves (for batches
                                   that leave SIMDs idle: the chain is bound by one wave's instruction issue rate) */
    int32_t n_spline_groups;    /* kind 3: descriptors in `spline` */
    int32_t waves_per_workgroup;/* rows_per_wave = 16 only: 1 (or 0), 2, 4, 8 independent waves per workgroup (each with its own
                                   rows and its own LDS region: LDS per workgroup = waves x tfep_inverse_block_lds_bytes_rows).
                                   Packs the launch onto fewer CUs, so that the look-ahead GEMMs of the next block -- whose
                                   waves need a SIMD's whole register file -- find empty CUs beside it. */
    int32_t paired;             /* rows_per_wave = 16 only: every workgroup is a PAIR of waves on the same 16 rows -- a consumer
                                   (the chain) and a loader that stages the next stage's weights and pre-activations into
                                   the other half of a double-buffered LDS stage (LDS: tfep_inverse_block_lds_bytes_paired) */
    /* SUPER-BLOCK launch (n_blocks > 0; rows_per_wave = 16, paired = 1): one launch runs the blocks blocks[0 .. n_blocks) one
     * after the other.  steps / feat_* / in_cols are then the concatenated tables of those blocks and every block has a record
     * of tfep_inverse_block_record_ints() int32:
     *   [0] n_steps, [1] first step record, [2] first entry of feat_cols / feat_sel / feat_in / feat_periodic (the step
     *   records' feat_off stay block-relative), [3] first entry of in_cols, [4 .. 7] cache_col0[l], [8 .. 11] cache_n_old[l],
     *   [12 + 4 l .. 15 + 4 l] for l = 0 .. n_layers (the last one = the output layer): (row0, n_rows, kb, ke) -- BEFORE the
     *   block's chain the pair adds, for its own 16 sample rows, the contribution of packed columns [kb, ke) of layer l - 1
     *   (l = 0: of the conditioner-input entries in_cols[kb .. ke), absolute positions in the concatenated table) to the
     *   packed rows [row0, row0 + n_rows) of layer l: exact-fp32 MFMA products of w[l] / wout with h[l - 1] / xpad, written
     *   to z_extra[l] / zout_extra (indexed like z[l] / zout) -- what earlier blocks of the same launch produced, i.e. what a
     *   caller of the one-block form supplies through short GEMMs between launches.  kb, ke multiples of 32 for l >= 1;
     *   [32] the number of features of the block (<= max_feats: the kernel keeps their indices, domains and y values in LDS).
     * waves_per_workgroup = 4 puts TWO pairs into a workgroup (2 x tfep_inverse_block_lds_bytes_paired of LDS): the same chains in
     * lockstep, the products' weight rows fetched once for both pairs' rows.
     * z_slabs[l] / zout_slabs then count the slabs of z[l] / zout WITHOUT the extra one (added per block when ke > kb);
     * cache_col0 and cache_n_old of the descriptor are ignored; n_steps = the most steps any block of the list has (1 .. max_feats
     * rounded up to 4: the kernel keeps the records of the block it works on -- [IB_BLK_INTS] + n_steps x [step_ints] -- in LDS). */
    int32_t n_blocks;
    const int32_t* blocks;
    float* z_extra[4];
    float* zout_extra;
    /* optional, super-block launches: the products of layer l = 1 .. n_layers (n_layers = the output layer) on split-f16 operands
     * -- ws[l]: the layer's split pack (tfep_masked_weight_prepare_split; rows in the order of w[l] / wout, row stride ldws[l]
     * floats), ws_inv_scale[l]: its 1/scale, h_inv_scale[l] (B): the per-row 1/scale fixed for the panel the layer reads
     * (h[l - 1]; a power of two, the one its split copy uses).  NULL: exact-fp32 products.  Index 0 is unused (layer 0 reads x). */
    const void* ws[5]; int64_t ldws[5];
    const float* ws_inv_scale[5];
    const float* h_inv_scale[5];
} tfep_inverse_block_desc;
int tfep_inverse_block_step_ints(void);
int tfep_inverse_block_record_ints(void);
/* LDS bytes a launch with these sizes needs (activation cache + input entries + the weight stage); a block fits iff
 * this is <= 160 KiB.  -1 for invalid arguments.  (_rows: for the given rows_per_wave; the plain form is 64.) */
int64_t tfep_inverse_block_lds_bytes(int n_layers, int cache_len, int max_feats);
int64_t tfep_inverse_block_lds_bytes_rows(int n_layers, int cache_len, int max_feats, int rows_per_wave);
int64_t tfep_inverse_block_lds_bytes_paired(int n_layers, int cache_len, int max_feats);
int tfep_inverse_block(const tfep_inverse_block_desc* desc, void* stream);
/* Diagnostics (TFEP_DIAG_INVERSE=1 in the library's environment; spline layers): cycles of the consumer wave of every pair of the
 * super-block launches so far, summed, in out[0..7] = {products at the head of a block, cache / table initialisation, waiting at a
 * hidden-layer hand-over, hidden dots, waiting at an output hand-over, parameter dot, transformer inverse + stores, whole kernel},
 * out[8] = pairs counted; clears the counters. */
int tfep_diag_inverse_cycles(unsigned long long* out);

/* ------------------------------------------------------------------------- */
/* One MAF layer in one launch (csrc/maf_layer.hip)                            */
/* ------------------------------------------------------------------------- */

/* AutoregressiveFlow.forward (flows/autoregressive.py:144-177) of a layer whose conditioner is a MADE of 1 .. 3 hidden
 * layers (conditioners/made.py:286-329, :355; masked.py:265-277) and whose transformer is the Moebius map of 2-vectors
 * (transformers/moebius.py:374-478), on split-f16 operands: every masked linear, ELU, the map and log|det J| for a
 * workgroup's 256 sample rows in ONE kernel -- for conditioners whose weights stay cache-resident (BASELINE cfg4-ii).
 *   a0 (B, lda0) / a0_inv_scale (B): the conditioner input as split rows (tfep_split_rows), lda0 a multiple of 32;
 *   w[l] (n_rows_w[l], ldw[l]) / w_scales[l] (4 floats) / bias[l]: the packed split weights of linear l
 *     (tfep_masked_weight_prepare_split: hidden units sorted by degree; the output layer in feature order), n_out[l] its
 *     real (padded to 32 for hidden layers) width, bias_absmax[l] (1 float, hidden layers) = max |bias|;
 *   k_ranges[l]: per column tile of tfep_maf_layer_tile_n() columns the [begin, end) of non-zero input columns
 *     (tfep_mask_k_ranges), multiples of 32;
 *   scratch[0], scratch[1] (B, ld_scratch): work panels for the hidden activations (split rows; every workgroup touches its
 *     own rows only), ld_scratch a multiple of 32 and >= every hidden width;
 *   kind 2 (the only one built): Moebius, moebius_dim = 2; x (B, ldx) fp32 is the map's input (feature c = column c of the
 *     output layer), y (B, ldy) and log_det_J (B) are written.  x, y: 8-byte aligned rows (even ldx / ldy).
 * Same arithmetic as tfep_split_rows + tfep_masked_linear_gemm (split, ELU, split_out) x hidden + (linear) +
 * tfep_moebius_forward: the accumulation order of a dot product and the row scales are those of the separate launches. */
typedef struct tfep_maf_layer_desc {
    int32_t B, n_linears;
    const void* a0; int64_t lda0; const float* a0_inv_scale;
    const void* w[4]; int64_t ldw[4]; int32_t n_rows_w[4]; int32_t n_out[4];
    const float* w_scales[4]; const float* bias[4]; const float* bias_absmax[4]; const int32_t* k_ranges[4];
    void* scratch[2]; int64_t ld_scratch;
    int32_t kind;
    const float* x; int64_t ldx; float* y; int64_t ldy; float* log_det_J;
    int32_t n_features, moebius_dim, moebius_unit_sphere; float moebius_max_radius;
} tfep_maf_layer_desc;
int tfep_maf_layer_tile_n(void);
int tfep_maf_layer_forward_split(const tfep_maf_layer_desc* desc, void* stream);
/* Diagnostics (TFEP_DIAG_MAF_LAYER=1 in the environment of the library): out[5] = cycles of wave 0 summed over the workgroups of
 * the launches so far in {k-loops, hidden epilogues, Moebius epilogues, whole kernel}, and the number of workgroups; clears them. */
int tfep_diag_maf_layer_cycles(unsigned long long* out);

/* ------------------------------------------------------------------------- */
/* Backward (training step, app/base.py:780-840 calls loss.backward())         */
/* ------------------------------------------------------------------------- */

/* out (C x R, row stride ld_out >= R) = in^T for in (R x C, row stride ld_in).  Layout helper of the
 * backward GEMMs: grad_input = g W needs W^T K-contiguous, grad_weight = g^T x needs g^T and x^T. */
int tfep_transpose(const float* in, int64_t ld_in, int R, int C, float* out, int64_t ld_out, void* stream);
/* out (C rows of split-f16 groups, row stride ld_out floats >= R_pad) = tfep_split_rows(in^T) in one pass (instead of
 * tfep_transpose + tfep_split_rows); rows r >= R of `in` read as zero, groups up to R_pad (a multiple of 8) written.
 *   mode 0: the per-tensor scale is known, scale_src[0] = 1 / scale (a matrix and its transpose share their maximum);
 *   mode 1: per-tensor scale computed here: inv_scale_out[0] = 1 / scale (inv_scale_out: 2 floats, [1] scratch);
 *   mode 2: one scale per output row from scale_src[c] = max_r |in[r, c]| (tfep_column_sums_absmax):
 *           inv_scale_out[c] = 1 / scale_c.
 * Same bits as the two-step form. */
int tfep_transpose_split(const float* in, int64_t ld_in, int R, int C, void* out_split, int64_t ld_out, int R_pad, int mode,
                         const float* scale_src, float* inv_scale_out, void* stream);
/* The transpose of split rows that share one scale (a packed weight matrix, tfep_masked_weight_prepare_split): out (C rows,
 * ld_out >= R) <- in (R rows, ld_in >= C), fp16 halves moved as they are (exact).  R, C multiples of 8.  The training
 * step gets W^T of grad_input = g W (masked.py:279-302) from the split pack of W without an fp32 copy of the matrix. */
int tfep_transpose_split_rows(const void* in_split, int64_t ld_in, int R, int C, void* out_split, int64_t ld_out, void* stream);
/* tfep_column_sums that also returns absmax[c] = max_r |in[r, c]| from the same pass. */
int tfep_column_sums_absmax(const float* in, int64_t ld, int R, int C, float* out, int accumulate, float* absmax, void* stream);
/* out[c] (+)= sum_r in[r, c]      grad_bias = grad_output.sum(0)  (masked.py:299-300) */
int tfep_column_sums(const float* in, int64_t ld, int R, int C, float* out, int accumulate, void* stream);
/* out[b, c] += in[b, c] */
int tfep_add_inplace(const float* in, int64_t ld_in, float* out, int64_t ld_out, int B, int C, void* stream);

/* Vector-Jacobian products of the transformers' FORWARD maps.  Given gy (B, D) = dL/dy and
 * g_log_det_J (B,) = dL/d log_det_J (may be NULL) they write gparams (layout glayout, every parameter
 * of every feature) and gx (B, D) = the DIRECT dL/dx (not through the conditioner).
 * Reference: eager autograd through affine.py:321-323 / spline.py:184-241, :319-417, :424-650.
 * The spline version covers every variant: circular, identity boundary slopes, learnable bounds. */
int tfep_affine_backward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout,
                         const float* gy, int64_t ldgy, const float* g_log_det_J,
                         float* gparams, tfep_param_layout glayout, float* gx, int64_t ldgx,
                         int B, int D, void* stream);
int tfep_spline_backward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout,
                         const tfep_spline_desc* desc, const float* gy, int64_t ldgy,
                         const float* g_log_det_J, float* gparams, tfep_param_layout glayout,
                         float* gx, int64_t ldgx, int B, int D, void* stream);

/* VJP of tfep_moebius_forward (moebius.py:374-478): gparams / gx (B, D), same `sign` as the forward. */
int tfep_moebius_backward(const float* x, int64_t ldx, const float* params, int64_t ldp,
                          int dimension, float max_radius, int unit_sphere, int sign,
                          const float* gy, int64_t ldgy, const float* g_log_det_J,
                          float* gparams, int64_t ldgp, float* gx, int64_t ldgx,
                          int B, int D, void* stream);
/* dst[b, c] = src[b, c] for a (B, C) block with row strides (VJP of the volume-preserving shift,
 * affine.py:366-411: gparams = gx = gy; and sub-blocks of MixedTransformer parameters). */
int tfep_copy_2d(const float* src, int64_t lds, float* dst, int64_t ldd, int B, int C, void* stream);

/* Gradient of the masked weight-norm parametrisation from the gradient of the PACKED effective weight
 * (as written by tfep_masked_weight_prepare with the same permutations):
 *   weight_g != NULL: grad_v (out, in), grad_g (out); masked entries of grad_v and fully-masked rows of
 *   grad_g are zero (the reference's gradient hooks, masked.py:401-402, :429);
 *   weight_g == NULL: grad_v = grad_weight = gW o mask (masked.py:293-297), grad_g unused. */
int tfep_weight_norm_backward(const float* gw_packed, int64_t ldw, const float* weight_v,
                              const float* weight_g, const float* mask, int out_features, int in_features,
                              const int32_t* row_of_out, const int32_t* col_of_in,
                              float* grad_v, float* grad_g, void* stream);
/* The same for a layer whose mask rows are prefixes of its packed columns (col_cut / in_of_col as in
 * tfep_masked_weight_prepare_prefix): the mask is not read, a row of weight_v goes through LDS once (in_features * 4 <=
 * 64 KiB) and the packed gradient is read over the live prefix only.  Same formulas; the double-precision sums run in
 * another (fixed) order. */
int tfep_weight_norm_backward_prefix(const float* gw_packed, int64_t ldw, const float* weight_v, const float* weight_g,
                                     int out_features, int in_features, const int32_t* row_of_out, const int32_t* in_of_col,
                                     const int32_t* col_cut, float* grad_v, float* grad_g, void* stream);

/* Backward of tfep_periodic_embedding: gx[:, f] for every input feature (periodic and not). */
int tfep_periodic_embedding_backward(const float* x, int64_t ldx, const int32_t* periodic_indices,
                                     int n_periodic, const int32_t* nonperiodic_indices, int n_nonperiodic,
                                     float lower, float upper, const float* gout, int64_t ldg,
                                     float* gx, int64_t ldgx, int B, void* stream);

/* Diagnostic (not on the path): run `iters` x 200 register-only v_mfma_f32_16x16x4_f32 per wave on
 * `blocks` workgroups of 512 threads -- the matrix-pipe ceiling of this device for the GEMM's own
 * instruction mix (2*25 accumulator tiles, 2 waves per SIMD).  scratch: blocks*512 floats.
 * flops = blocks * 8 waves * iters * 200 * 2048. */
int tfep_diag_mfma_peak(float* scratch, int blocks, int iters, void* stream);

/* ------------------------------------------------------------------------- */
/* TFEP reductions (tfep/loss.py, tfep/analysis/estimator.py)                 */
/* ------------------------------------------------------------------------- */

/*
 * One pass over the local shard producing the sufficient statistics of
 * BoltzmannKLDivLoss.forward (loss.py:125-140) and fep_estimator (estimator.py:73-86):
 *   r_i = u_B[i] - ldj[i] - u_A[i]                (ldj / u_A may be NULL)
 *   out[0] = count of non-NaN r_i (or all, if !ignore_nan)     out[1] = sum r_i
 *   out[2] = m_w = max_i log_w[i]       out[3] = sum exp(log_w - m_w)
 *   out[4] = sum exp(log_w - m_w) r_i                           (log_w may be NULL)
 *   out[5] = m_e = max_i (-r_i/kT + bias_i/kT)    out[6] = sum exp(-r_i/kT + bias_i/kT - m_e)
 *   out[7] = m_b = max_i bias_i/kT      out[8] = sum exp(bias_i/kT - m_b)   (bias may be NULL)
 * out is (9) float64 on the device; workspace: tfep_tfep_reduce_workspace_doubles(N) float64.  Statistics combine across shards / GPUs by
 * max + rescaled sum (one all-reduce of <= 9 scalars).
 */
int tfep_tfep_reduce(const float* target_potentials, const float* log_det_J,
                     const float* ref_potentials, const float* log_weights, const float* bias,
                     float kT, int ignore_nan, int N, double* workspace, double* out, void* stream);
/* Size (in doubles) of the device workspace tfep_tfep_reduce needs for N samples. */
int tfep_tfep_reduce_workspace_doubles(int N);

/*
 * Bootstrap distribution of fep_estimator (analysis/bootstrap.py:185-262 with statistic = fep_estimator,
 * estimator.py:73-86), one workgroup per resample, the resampled data never materialised:
 *   out[r] = -kT (logsumexp_j(-work[i]/kT [+ bias[i]/kT]) - log sample_size [or - logsumexp_j bias[i]/kT]),
 *   i = indices[r][j]  (indices: (n_resamples, sample_size) int64 in [0, n_data), as drawn by torch.randint), or
 *   Bayesian bootstrap (indices == NULL, i = j, sample_size <= n_data): weights (n_resamples, sample_size) summing to
 *   one per row, out[r] = -kT logsumexp_j(-work[j]/kT + log weights[r][j])   (bias must be NULL).
 * work / bias: (n_data) fp32; out: (n_resamples) float64.
 */
int tfep_bootstrap_fep(const float* work, const float* bias, const int64_t* indices, const float* weights,
                       int64_t n_data, int64_t n_resamples, int64_t sample_size, float kT, double* out, void* stream);

/* ------------------------------------------------------------------------- */
/* Config 5: EGNN dynamics of the continuous flow (tfep/nn/dynamics/egnn.py,   */
/* tfep/nn/graph.py, tfep/nn/embeddings/radial.py, tfep/nn/flows/continuous.py) */
/* ------------------------------------------------------------------------- */

/*
 * Tile count nt (feature tile = 16 nt) of the fused EGNN kernels for a (node_feat_dim, distance_feat_dim) pair:
 * 1, 2 or 4; 0 = unsupported (a dimension above 64).  Node-level arrays (h, P, Q, aggregated messages) are
 * (B, n_nodes, 16 nt) float32, zero in the padding.
 */
int tfep_egnn_tile(int node_feat_dim, int distance_feat_dim);
/* Floats of one packed layer (tfep_egnn_pack_layer) for a tile count; -1 if nt is not 1, 2 or 4. */
int64_t tfep_egnn_packed_floats(int nt);

/* The parameter tensors of one _EGLayer (egnn.py:225-270), reference shapes, row-major. */
typedef struct tfep_egnn_layer_params {
    int32_t F, G;                   /* node_feat_dim, distance_feat_dim */
    const float* dist_means;        /* (G)  distance_embedding._means = linspace(0, r_cutoff, G)  (radial.py:132-137) */
    const float* dist_log_gammas;   /* (G)  distance_embedding._log_gammas                        (radial.py:57-62) */
    const float* msg0_w;            /* (F, 2F+G) message_mlp.0.weight: columns [h_src | h_dest | rbf] (egnn.py:312) */
    const float* msg0_b;            /* (F) */
    const float* msg2_w;            /* (F, F)  message_mlp.2 */
    const float* msg2_b;            /* (F) */
    const float* att_w;             /* (1, F)  attention_mlp.0 */
    const float* att_b;             /* (1) */
    const float* ux0_w;             /* (F, F)  update_x_mlp.0 */
    const float* ux0_b;             /* (F) */
    const float* ux2_w;             /* (1, F)  update_x_mlp.2 (no bias) */
    const float* uh0_w;             /* (F, 2F) update_h_mlp.0: columns [h | aggregated messages] (egnn.py:336) */
    const float* uh0_b;             /* (F) */
    const float* uh2_w;             /* (F, F)  update_h_mlp.2 */
    const float* uh2_b;             /* (F) */
} tfep_egnn_layer_params;

/*
 * Re-pack the parameters of one layer for the kernels below (every forward, like the masked weights): the six F x F
 * blocks as "lane-linear" MFMA operand images (csrc/egnn.hip), the vectors zero-padded to 16 nt, exp(log_gamma).
 * packed: tfep_egnn_packed_floats(nt) floats.
 */
int tfep_egnn_pack_layer(const tfep_egnn_layer_params* params, int nt, float* packed, void* stream);

/*
 * EGNNDynamics._create_node_embedding (egnn.py:196-219) and the node-level part of the first layer's message MLP:
 *   h0[i] = W_emb [one_hot(type_i), exp(-exp(log_gamma_k) (t - mean_k)^2)] + b_emb      (the same for every sample)
 *   P0[i] = W1[:, 0:F] h0[i],   Q0[i] = W1[:, F:2F] h0[i] + b1                           (message_mlp.0 split by input)
 * one_hot: (n_nodes, n_types); w_emb: (F, n_types + time_dim); h0, P0, Q0: (n_nodes, 16 nt).
 */
int tfep_egnn_embed(const float* one_hot, int n_nodes, int n_types, float t, const float* time_means,
                    const float* time_log_gammas, int time_dim, const float* w_emb, const float* b_emb,
                    const tfep_egnn_layer_params* layer0, int nt, float* h0, float* P0, float* Q0, void* stream);

/*
 * One _EGLayer.forward without its node MLP (egnn.py:272-369): for every sample and every destination node j
 *   nm[j]      = sum_{i != j, |x_j - x_i| <= r_cutoff} m_ij              (_create_edge_messages + segment sum, :294-331)
 *   pos_out[j] = pos[j] + speed_factor sum_i (x_j - x_i)/|x_j - x_i| tanh(x2 . SiLU(X1 m_ij + d1))      (:344-364)
 *   m_ij = m2 sigmoid(wa . m2 + ba),  m2 = SiLU(W2 SiLU(P[i] + Q[j] + W1c rbf(|x_j - x_i|)) + b2)
 * with the Behler-Parrinello radial basis rbf (radial.py:269-291).  No edge list, no scatter_add (graph.py:119-316):
 * see csrc/egnn.hip.  With dpos != NULL the directional derivative along (dpos, dP, dQ) is carried through the same
 * pass: dpos_out, dnm.  P, Q: (B, n_nodes, 16 nt) with pq_bstride = n_nodes, or (n_nodes, 16 nt) shared by all samples
 * with pq_bstride = 0 (layer 0); dP / dQ NULL = zero (layer 0).  nm NULL: the aggregated messages are not needed (last
 * layer: the velocity depends on the positions only).
 */
typedef struct tfep_egnn_edge_args {
    int32_t B, n_nodes, nt;
    int32_t split;              /* != 0: the three per-edge F x F products on split-f16 operands (3 fp16 MFMAs per fp32
                                   product, fp32 accumulate; csrc/egnn.hip), 0: exact-fp32 MFMA */
    float r_cutoff, speed_factor;
    const float* packed;
    const float* pos;
    const float* dpos;
    const float* P;
    const float* Q;
    int64_t pq_bstride;
    const float* dP;
    const float* dQ;
    float* pos_out;
    float* dpos_out;
    float* nm;
    float* dnm;
} tfep_egnn_edge_args;
int tfep_egnn_edge(const tfep_egnn_edge_args* args, void* stream);

/*
 * Node update between layer l and l + 1: h' = h + U2 SiLU(U1 [h, nm] + c1) + c2 (_update_h, egnn.py:327-342) with the
 * packed weights of layer l, then P' = W1[:, 0:F] h', Q' = W1[:, F:2F] h' + b1 with those of layer l + 1; tangents
 * alongside when dnm != NULL (dh NULL = zero).  h: (B, n_nodes, 16 nt) with h_bstride = n_nodes or shared, 0.
 */
typedef struct tfep_egnn_node_args {
    int32_t B, n_nodes, nt;
    const float* packed;
    const float* packed_next;
    const float* h;
    int64_t h_bstride;
    const float* dh;
    const float* nm;
    const float* dnm;
    float* h_out;
    float* dh_out;
    float* P_out;
    float* Q_out;
    float* dP_out;
    float* dQ_out;
} tfep_egnn_node_args;
int tfep_egnn_node(const tfep_egnn_node_args* args, void* stream);

/*
 * Reverse pass (vector-Jacobian product with respect to the positions) of tfep_egnn_edge: what the reference obtains by
 * torch.autograd.grad(vel, x, e) in its Hutchinson estimators (continuous.py:307-361).  The layer is recomputed per edge
 * and walked backwards; run it TWICE per layer: src_owned = 0 (workgroups own destinations) writes
 *   g_lane = g_Q (B, n, 16 nt)  and  g_pos = g_pos_out + destination-side position terms,
 * then src_owned = 1 (workgroups own sources) writes g_lane = g_P and ADDS the source-side terms to g_pos.
 * pos, P, Q: the layer's inputs (saved from the forward pass); g_pos_out: cotangent of its output positions;
 * g_nm: cotangent of its aggregated messages (from tfep_egnn_node_backward; NULL = zero: last layer).
 */
typedef struct tfep_egnn_edge_bwd_args {
    int32_t B, n_nodes, nt, split, src_owned;
    float r_cutoff, speed_factor;
    const float* packed;
    const float* pos;
    const float* P;
    const float* Q;
    int64_t pq_bstride;
    const float* g_pos_out;
    const float* g_nm;
    float* g_lane;
    float* g_pos;
} tfep_egnn_edge_bwd_args;
int tfep_egnn_edge_backward(const tfep_egnn_edge_bwd_args* args, void* stream);

/*
 * Reverse pass of tfep_egnn_node (same packed / packed_next): from the cotangents of its outputs (g_h_next: of h', NULL =
 * zero; g_P, g_Q: of the next layer's source / destination terms) to those of its inputs, g_h and g_nm; h, nm: the
 * node kernel's inputs saved from the forward pass.
 */
typedef struct tfep_egnn_node_bwd_args {
    int32_t B, n_nodes, nt;
    const float* packed;
    const float* packed_next;
    const float* h;
    int64_t h_bstride;
    const float* nm;
    const float* g_h_next;
    const float* g_P;
    const float* g_Q;
    float* g_h;
    float* g_nm;
} tfep_egnn_node_bwd_args;
int tfep_egnn_node_backward(const tfep_egnn_node_bwd_args* args, void* stream);

/* out = sign (in - mean over the nodes), per sample and component, (B, 3 n_nodes): the centring of the velocity
 * (egnn.py:187-191) and, being a symmetric projector, its own reverse pass. */
int tfep_egnn_center(const float* in, int B, int n_nodes, float sign, float* out, void* stream);

/* Per row of (B, D): dot[b] += scale x[b] . y[b], sumsq[b] += scale |x[b]|^2 (either may be NULL): the trace and Frobenius
 * estimates of continuous.py:307-361 from x = e^T J, y = e. */
int tfep_row_dots(const float* x, const float* y, int B, int D, float scale, float* dot, float* sumsq, void* stream);

/*
 * End of EGNNDynamics.forward (egnn.py:178-193): vel = (pos - x) - mean_nodes(pos - x), (B, 3 n_nodes).  With a tangent
 * (dpos, eps): jvp = (dpos - eps) - mean = J eps, and the quadratic forms of the trace estimators
 * (continuous.py:285-324):  trace[b] += scale eps . (J eps),  frob[b] += scale |J eps|^2.  vel_sq[b] = |vel|^2
 * (continuous.py:281-282).  vel, jvp, trace, frob, vel_sq may be NULL.
 */
int tfep_egnn_finish(const float* pos, const float* x, const float* dpos, const float* eps, int B, int n_nodes,
                     float* vel, float* jvp, float scale, float* trace, float* frob, float* vel_sq, void* stream);

/*
 * GaussianBasisExpansion.forward (radial.py:110-130): out[e, k] = exp(-exp(log_gamma_k) (r_e - mean_k)^2); with
 * switching != 0 times the Behler-Parrinello cosine switch 0.5 cos(pi r / r_cutoff) + 0.5, zero beyond the cutoff if
 * force_zero_after_cutoff (radial.py:161-176, 269-291).  r: (n); out: (n, n_basis).
 */
int tfep_radial_expansion(const float* r, int64_t n, const float* means, const float* log_gammas, int n_basis,
                          float r_cutoff, int switching, int force_zero_after_cutoff, float* out, void* stream);

/* unsorted_segment_sum (graph.py:304-316): out[s, :] = sum_{rows with segment_ids == s} data[row, :]; out is cleared
 * first; float atomics (the fused EGNN kernels do not use this). */
int tfep_segment_sum(const float* data, const int64_t* segment_ids, int64_t n_rows, int n_cols, int64_t n_segments,
                     float* out, void* stream);

/* y = x + sum_{k < n_terms} a[k] v[k]  (x may be NULL = 0; n_terms <= 4; v, a: HOST arrays of device pointers /
 * coefficients): the stage combinations of the fixed-grid ODE steppers that replace torchdiffeq's
 * (continuous.py:136-169). */
int tfep_ode_axpy(const float* x, const float* const* v, const float* a, int n_terms, int64_t n, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFEP_HIP_H */
