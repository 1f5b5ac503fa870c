/* tdeed_hip.h -- C ABI of libtdeed_hip.so, the gfx950 (MI355X) kernels of the T-DEED hot path.
 *
 * The reference (arturxe2/T-DEED) has no FFI/plugin layer: its "operator interface" for this
 * path is the set of torch.nn module calls inside TDEEDModel.Impl.forward
 * (/root/reference/model/model.py:105-149) and the loss in TDEEDModel.epoch (model.py:308-319).
 * Each entry point below replaces one fused group of those calls; the cited file:line is what
 * it replaces.  Conventions:
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (hipMalloc /
 *     torch tensor .data_ptr()); the library allocates nothing and never synchronises;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered on it;
 *   - `dtype` selects the storage type of activations and dense weights: TDEED_F32 (parity
 *     mode, exact-f32 MFMA / VALU) or TDEED_BF16 (throughput mode, fp32 accumulation);
 *     parameters documented as `float*` are always fp32;
 *   - activations are channels-last: images NHWC ([frame][y][x][c]), sequences NTC ([clip][t][c]);
 *   - return 0 on success, <0 on error; tdeed_last_error() gives the message (thread-local).
 * Channel counts must be multiples of 8 (true for every RegNetY-200MF/800MF and SGP width).
 */
#ifndef TDEED_HIP_H
#define TDEED_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { TDEED_F32 = 0, TDEED_BF16 = 1 };
enum { TDEED_ACT_NONE = 0, TDEED_ACT_RELU = 1, TDEED_ACT_GELU = 2 };
enum { TDEED_OK = 0, TDEED_ERR_ARG = -1, TDEED_ERR_LAUNCH = -2, TDEED_ERR_RUNTIME = -3 };

const char* tdeed_last_error(void);
int tdeed_abi_version(void);
/* Device sanity: returns 0 and fills name (<=63 chars) / CU count when device `dev` is gfx950. */
int tdeed_device_info(int dev, char* name64, int* n_cu, int* is_gfx950);

/* ---- pre-proc + stem -------------------------------------------------------------------
 * normalize /255, crop, optional h-flip, ImageNet standardize (model.py:107-129,151-167) fused
 * into timm RegNet stem Conv3x3 s2 (3->32) + BN(eval, folded to scale/shift) + ReLU (model.py:133).
 * frames: uint8 [N][3][H][W] (NCHW as the loader delivers it); out: [N][Ho][Wo][32], Ho=(crop_h+1)/2. */
int tdeed_stem_fwd(const void* frames /* uint8, or fp32 0..255 when frames_f32 (mixup batches) */, int frames_f32,
                   int N, int H, int W, int crop_top, int crop_left,
                   int crop_h, int crop_w, int flip /* all frames (eval TTA, model.py:159-162) */,
                   const unsigned char* flip_mask /* NULL, or [N] per-frame h-flip flags: the train-time
                                                     RandomHorizontalFlip drawn per clip (model.py:83,154-157) */,
                   const float* w /*[32][3][3][3]*/,
                   const float* scale /*[32]*/, const float* shift /*[32]*/, void* out,
                   int relu /* 0: raw conv*scale+shift (training) */, int dtype, void* stream);

/* Training stem (bf16): the same pre-processing + conv on the MFMA pipe, RAW conv output z [N][Ho][Wo][32] (BatchNorm runs
 * on batch statistics in training, model.py:133 under .train()) and per workgroup the per-channel sum / sum of squares of what
 * it stored: colpart fp32 [N * tdeed_stem_mfma_parts(crop_h, crop_w)][2][32] (fold with tdeed_bn_finalize(colpart,
 * colpart + 32, 64, N * parts, N*Ho*Wo, 32, ...)).  wfrag: the [2][2][64][8] fragments of tdeed_s1_front_fwd's stem_wf in fp32 (operands are split into bf16 head + tail
 * in the kernel: three MFMAs per product, z is the fp32 conv rounded once).
 * frames / crop / flip / flip_mask as tdeed_stem_fwd.  tdeed_stem_mfma_parts == 0: the row band does not fit LDS. */
int tdeed_stem_mfma_parts(int crop_h, int crop_w);
int tdeed_stem_mfma_fwd(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left, int crop_h,
                        int crop_w, int flip, const unsigned char* flip_mask, const void* wfrag, void* z, float* colpart,
                        void* stream);

/* ---- fused trunk front (bf16 only): pre-proc + stem + s1.b1.conv1 + s1.b1.conv2 (+SE squeeze) + s1.b1.downsample
 * uint8 frames [N][3][H][W] -> y2 [N][Ho][Wo][C1] (conv2 output), shortcut [N][Ho][Wo][C1], pooled fp32
 * [N][parts][C1] partial sums of y2 (parts = tdeed_s1_front_parts(crop_h, crop_w, C1)).  The 112^2 stem and conv1
 * maps stay in LDS / MFMA accumulators (front.hip: a workgroup walks a strip of output rows over rings of input and
 * conv1 rows; parts = strips per frame, a function of the shape and of TDEED_FRONT_ROLL / TDEED_FRONT_PIPE only, never of
 * the frames' alignment).  Weight fragments are pre-packed by
 * tdeed_amd.engine.pack_front_weights; scale/shift are the folded eval BatchNorms (fp32). */
int tdeed_s1_front_parts(int crop_h, int crop_w, int C1);
int tdeed_s1_front_fwd(const uint8_t* frames, int N, int H, int W, int crop_top, int crop_left, int crop_h,
                       int crop_w, int flip, const void* stem_wf, const float* stem_sc, const float* stem_sh,
                       int C1, const void* w1f, const float* sc1, const float* sh1, const void* wdf,
                       const float* scd, const float* shd, const void* w2f, const float* sc2,
                       const float* sh2, void* y2, void* shortcut, float* pooled, void* stream);

/* ---- dense 1x1 contraction (MFMA) ---------------------------------------------------------
 * C[m][n] = act( (sum_k A'[m][k] * W[n][k]) * scale[n] + shift[n] + R[m][n] )
 * Replaces every 1x1 Conv2d+BN(eval)(+ReLU) of the RegNetY trunk incl. the stride-2 shortcut,
 * the SE re-scale in front of conv3, the residual add + ReLU (timm Bottleneck, model.py:133),
 * GatedShift's channel splice (shift.py:89-93) and the SGP mlp / concat_fc Conv1d(k=1)
 * (modules.py:134-138,186,248-254,308-309,316).
 *   A   [M][K] row stride lda (elements).  If A0 != NULL, columns [0,k0) are read from A0
 *       (row stride lda0) instead: the gate-shift output spliced in front of the pass-through
 *       channels.  If a_scale != NULL: A'[m][k] = A[m][k] * a_scale[(m / a_scale_rows)*K + k]
 *       (SE gate per frame).  If gather_stride > 1: row m = (f, yo, xo) of an
 *       [F][gather_ho][gather_wo] grid reads input row (f, yo*s, xo*s) of [F][gather_hi][gather_wi].
 *   W   [N][K] row stride ldw, same dtype as A.   scale/shift: fp32 [N], either may be NULL.
 *   R   optional residual [M][N] row stride ldr.   C: [M][N] row stride ldc.
 * K, N, k0, lda, lda0, ldw, ldr, ldc must be multiples of 8.
 *   colpart  optional fp32 [ceil(M/128)][2][N]: per 128-row tile the column sums and sums of squares of the stored C
 *       (training: the BatchNorm batch statistics of a raw conv output come out of the conv's own epilogue; fold them
 *       with tdeed_bn_finalize(colpart, colpart + N, 2N, ceil(M/128), M, N, ...)).
 *   C2  optional second output [M][n2] row stride ldc2 (n2, ldc2 multiples of 8, n2 <= N): a compact copy of columns
 *       [0, n2) of C.  The next bottleneck's gate-shift reads only that channel slice (shift.py:46-93), three times; out of
 *       the channels-last map every such read drags whole rows' cache lines along (measured 4.5x the slice's bytes).
 *       c2_pre (tdeed_gemm_fwd only): C2 receives the value before residual and activation and columns [0, n2) of C keep the
 *       residual alone -- the input gradient of a gate-shifted conv1 (shift.py:89-93 backwards): those columns belong to the
 *       gate-shift module's backward, the block's shortcut gradient R joins the rest in this epilogue. */
int tdeed_gemm_fwd(const void* A, long lda, const void* A0, long lda0, int k0,
                   const float* a_scale, int a_scale_rows, int M, int K, int N, const void* W,
                   long ldw, const float* scale, const float* shift, const void* R, long ldr,
                   int act, void* C, long ldc, int gather_stride, int gather_hi, int gather_wi,
                   int gather_ho, int gather_wo, float* colpart, void* C2, long ldc2, int n2, int c2_pre, int dtype,
                   void* stream);

/* Weight-stationary variant of the same contraction for narrow layers (whole W in LDS, activations
 * streamed global->registers in MFMA fragment shape, persistent blocks; see gemm.hip).  Same
 * semantics and arguments as tdeed_gemm_fwd except that the weights arrive pre-packed:
 * Wfrag = tdeed_amd.engine.pack_ws_weights(W): [2*ceil(N/32)][ceil(K/(4*epc))][64 lanes][16 B], rows
 * permuted so that each lane owns 8 consecutive output channels.  tdeed_gemm_ws_fits() tells whether
 * (K, N, dtype) fits (LDS budget 64 KB). */
int tdeed_gemm_ws_fits(int K, int N, int dtype);
int tdeed_gemm_ws_fwd(const void* A, long lda, const void* A0, long lda0, int k0,
                      const float* a_scale, int a_scale_rows, int M, int K, int N, const void* Wfrag,
                      const float* scale, const float* shift, const void* R, long ldr, int act,
                      void* C, long ldc, int gather_stride, int gather_hi, int gather_wi,
                      int gather_ho, int gather_wo, void* C2, long ldc2, int n2, int dtype, void* stream);

/* ---- grouped 3x3 conv + BN + ReLU + SE squeeze ---------------------------------------------
 * timm Bottleneck.conv2 (groups = C/gw, stride 1|2, pad 1) + BN(eval) + ReLU, and the SE
 * squeeze (sum over H,W) of its output.  x: [N][Hi][Wi][C], y: [N][Ho][Wo][C], gw in {8,16}.
 *   w      fp32 [G][9][gw_in][gw_out] (tap-major repack of Conv2d.weight [C][gw][3][3]); used by the
 *          VALU kernel (TDEED_F32, or bf16 when wfrag is NULL).
 *   wfrag  bf16 MFMA operand fragments [ceil4(C/16)][5][64][8] (see tdeed_amd.engine.pack_gconv_frags);
 *          TDEED_BF16 only: implicit GEMM on v_mfma_f32_16x16x32_bf16 from an LDS-staged halo band.
 *   pooled fp32 [N][parts][C]: per-band partial SUMS of y over pixels, parts =
 *          tdeed_gconv3x3_parts(Hi, Wi, C, stride, dtype) (1 on the VALU path).
 *   pooled_sq  optional (bf16 MFMA path), same shape: partial sums of y^2 -- with pooled the BatchNorm batch statistics
 *          of the raw conv output (tdeed_bn_finalize(pooled, pooled_sq, C, N*parts, N*Ho*Wo, C, ...)). */
int tdeed_gconv3x3_parts(int Hi, int Wi, int C, int stride, int dtype);
int tdeed_gconv3x3_mfma_fits(int Hi, int Wi, int C, int stride);   /* 1: the bf16 MFMA kernel (wfrag, in_a) serves it */
/*   in_a, in_b  optional fp32 [C] (bf16 MFMA path, training): x is the RAW output of the conv in front and
 *          relu(in_a[c] * x + in_b[c]) -- the BatchNorm(batch statistics) + ReLU between the two convs -- is applied while
 *          the band is staged, so the post-BN map is never written; bit-identical to running on the materialised map. */
int tdeed_gconv3x3_fwd(const void* x, int N, int Hi, int Wi, int C, int gw, int stride,
                       const float* w, const void* wfrag, const float* scale, const float* shift,
                       void* y, float* pooled, float* pooled_sq, const float* in_a, const float* in_b,
                       int relu /* 0: y = conv*scale+shift (training: raw map for the batch statistics) */, int dtype,
                       void* stream);

/* Split-K form of the contraction for the short sequences of the SGP encoder-decoder (a few hundred rows, K up to
 * 6C; bf16): C = act((A . W^T) * scale + shift + R).  Two launches: S = tdeed_gemm_splitk_splits(K) partial products
 * into `workspace` (fp32, S*M*N elements, caller-owned), then a reduce + epilogue pass. */
int tdeed_gemm_splitk_splits(int K);
int tdeed_gemm_splitk_partials(const void* A, long lda, int M, int K, int N, const void* W, long ldw, float* workspace,
                               void* stream);   /* the partial products only: workspace [splits(K)][M][N] fp32 */
int tdeed_gemm_splitk_fwd(const void* A, long lda, int M, int K, int N, const void* W, long ldw, const float* scale,
                          const float* shift, const void* R, long ldr, int act, void* C, long ldc,
                          float* workspace, void* stream);

/* SE excitation on the MFMA pipe (bf16 weights as A-operand fragments [ceil(R/16)][ceil(C/32)][64][8] and
 * [ceil(C/16)][ceil(R/32)][64][8], tdeed_amd.engine.pack_se_mfma); 16 frames per workgroup; same contract as
 * tdeed_se_gate_bf16_fwd.  tdeed_se_gate_mfma_fits(C, R) != 0 tells whether the shape is covered (C <= 384, R <= 96). */
int tdeed_se_gate_mfma_fits(int C, int R);
int tdeed_se_gate_mfma_fwd(const float* pooled, int n_parts, float inv_cnt, int N, int C, int R, const void* w1f,
                           const float* b1, const void* w2f, const float* b2, float* gate, void* stream);

/* Register-stationary form of the same contraction for K = N = 320 over many rows (conv1 / conv3 of the s3 blocks of
 * RegNetY-800MF): the whole weight matrix lives in the registers of a 10-wave workgroup, the activations cross the chip once
 * (64-row tiles through LDS).  Arguments as tdeed_gemm_ws_fwd (Wfrag from pack_ws_weights, bf16) without the row gather;
 * a_scale_rows >= 64.  tdeed_gemm_rs_fits(M, K, N) != 0 tells whether the shape is covered. */
int tdeed_gemm_rs_fits(int M, int K, int N);
int tdeed_gemm_rs_fwd(const void* A, long lda, const void* A0, long lda0, int k0, const float* a_scale, int a_scale_rows,
                      int M, int K, int N, const void* Wfrag, const float* scale, const float* shift, const void* R, long ldr,
                      int act, void* C, long ldc, void* C2, long ldc2, int n2, void* stream);

/* the same contraction as the training forward runs it: raw output + the BatchNorm statistics of that output, one partial
 * row per persistent workgroup: colpart fp32 [tdeed_gemm_rs_grid(M)][2][N] (column sums | sums of squares) */
int tdeed_gemm_rs_grid(int M);
int tdeed_gemm_rs_stats_fwd(const void* A, long lda, const void* A0, long lda0, int k0, int M, int K, int N, const void* Wfrag,
                            void* C, long ldc, float* colpart, void* stream);

/* conv1 (1x1 + BN + ReLU, with the gate-shift splice of shift.py:89-93) IN FRONT of the grouped 3x3 of the same timm
 * Bottleneck, one launch: the y1 band the grouped conv reads is computed in LDS from the block input, the y1 map (the
 * largest intermediate of a stride-2 block) never exists.  bf16.  x [N][Hi][Wi][Cin]; G optional [N*Hi*Wi][Fp]; w1f: conv1
 * weight [C][Cin] as MFMA A-operand fragments [tdeed_c1_gconv_slab_tiles()][ceil(Cin/32)][64][8] (zero padded to whole
 * slabs); s1 / h1: its folded BatchNorm; wfrag / scale / shift / y / pooled as tdeed_gconv3x3_fwd (ReLU applied).
 * Bit-identical to tdeed_gemm_fwd (or tdeed_gemm_ws_fwd) followed by tdeed_gconv3x3_fwd.  tdeed_c1_gconv_fits: Cin <= 64, or 97..160 (k-steps of 32: 1, 2, 4, 5). */
/* diagnostic: int64 [workgroups][8] phase time stamps of tdeed_c1_gconv_fwd (wall_clock64, 10 ns ticks; 0 start, 1 weights
 * requested + halo zeroed, 2 wave 0's conv1 tiles done, 3 barrier, 4 grouped conv set up, 5 its tiles multiplied and stored,
 * 6 squeeze sums stored); null switches it off */
int tdeed_c1_gconv_set_debug(void* buf);
int tdeed_c1_gconv_fits(int Hi, int Wi, int Cin, int C, int stride);
int tdeed_c1_gconv_slab_tiles(int Hi, int Wi, int C, int stride);
int tdeed_c1_gconv_fwd(const void* x, const void* G, int Fp, int N, int Hi, int Wi, int Cin, int C, int gw, int stride,
                       const void* w1f, const float* s1, const float* h1, const void* wfrag, const float* scale,
                       const float* shift, void* y, float* pooled, void* stream);

/* A whole stride-1 RegNetY bottleneck with identity shortcut on a small map in ONE launch (timm Bottleneck.forward:
 * conv1 -> conv2 -> se -> conv3 + shortcut -> ReLU, with the gate-shift splice of shift.py:89-93 on conv1's operand;
 * SURVEY §8 a2 / a3): the frames of a workgroup stay in LDS, only x and the output cross HBM.  bf16.
 *   x [N][h][w][C]; G optional [N*h*w][Fp] compact gate-shift output replacing channels [0, Fp) of conv1's operand (the
 *   residual is x itself); w1f / w3f: conv weights [C][C] as MFMA A-operand fragments [ceil(C/16)][ceil(C/32)][64][8]
 *   (tdeed_amd.engine.pack_mfma_frags); w2f as for tdeed_gconv3x3_fwd (pack_gconv_frags, group width 8 or 16); se_w1f /
 *   se_w2f / R as for tdeed_se_gate_mfma_fwd; s*, h*: folded BatchNorm scale / shift; out [N][h][w][C]; out2 optional
 *   [N*h*w][n2] compact copy of channels [0, n2) (the next block's gate-shift slice).
 * Bit-identical to tdeed_gemm_fwd -> tdeed_gconv3x3_fwd -> tdeed_se_gate_mfma_fwd -> tdeed_gemm_fwd on the same operands.
 * tdeed_bneck_fits: 7x7x368 (two frames per workgroup) and 14x14x152 are the shapes it was built for.
 * tdeed_bneck_set_debug(buf): diagnostic, int64 [workgroups][16] phase time stamps (null switches it off). */
int tdeed_bneck_fits(int h, int w, int C, int R);
int tdeed_bneck_set_debug(void* buf);
int tdeed_bneck_fwd(const void* x, const void* G, int Fp, int N, int h, int w, int C, const void* w1f, const float* s1,
                    const float* h1, const void* w2f, const float* s2, const float* h2, const void* se_w1f,
                    const float* se_b1, const void* se_w2f, const float* se_b2, int R, const void* w3f, const float* s3,
                    const float* h3, void* out, void* out2, int n2,
                    int w2_tap_major /* k-slot order of w2f: 1 = engine.pack_gconv_frags(tap_major=True), the conflict-free
                                        order of this launch (group width 8); 0 = the order tdeed_gconv3x3_fwd reads */,
                    void* stream);
/* The same block behind a gate-shift-fuse site (impl/gsf.py:74-93) with the site's last launch -- fusion weights + blend,
 * tdeed_gsf_blend_src_fwd -- done inside the frame load: gx [N][h*w][ldx] is the slice's source (the block input, or the
 * compact copy of its first Fp channels), gate / ysum / xsum are tdeed_gsf_gate_fwd's outputs for the N = B * T frames, cw* / cb*
 * the two fusion convs.  out == tdeed_bneck_fwd(x, G = tdeed_gsf_blend_src_fwd(gx, ...)), bit for bit.
 * Q (optional, tdeed_bneck_qtail_fits): the tap maps [N][h*w][6] of the NEXT block's gate-shift site (impl/gsf.py:49-52, the
 * conv3d as three 2-D convs per frame), made from this block's output rows while they are in LDS -- what the first launch of
 * tdeed_gsf_gate_fwd computes from `out`, as a 1x1 contraction to per-tap sums plus nine fp32 adds (same products, another
 * summation order: equal to fp32 rounding); the site then runs tdeed_gsf_gate_sums_fwd only.  q_wqf: the site's weights as
 * engine.pack_gsf_p_frags lays them out; q_bn [2][8 * ceil(q_F / 8)]: its folded BatchNorm3d scale | shift, zeros behind
 * channel q_F. */
int tdeed_bneck_qtail_fits(int h, int w, int C, int F);
int tdeed_bneck_gs_fwd(const void* x, const void* gx, int ldx, const float* gate, const float* ysum, const float* xsum,
                       const float* cw1, const float* cb1, const float* cw2, const float* cb2, int T, int F, int Fp, int N,
                       int h, int w, int C, const void* w1f, const float* s1, const float* h1, const void* w2f,
                       const float* s2, const float* h2, const void* se_w1f, const float* se_b1, const void* se_w2f,
                       const float* se_b2, int R, const void* w3f, const float* s3, const float* h3, void* out, void* out2,
                       int n2, int w2_tap_major, const void* q_wqf, const float* q_bn, int q_F, float* Q, void* stream);


/* ---- SE excitation: gate = sigmoid(W2 relu(W1 mean + b1) + b2) -----------------------------
 * timm SEModule fc1/ReLU/fc2/sigmoid.  pooled: fp32 [N][n_parts][C] partial sums, mean = inv_cnt *
 * sum over parts; gate: fp32 [N][C]; w1t [C][R] (fc1.weight transposed), w2t [R][C] (fc2.weight transposed). */
int tdeed_se_gate_fwd(const float* pooled, int n_parts, float inv_cnt, int N, int C, int R,
                      const float* w1t, const float* b1, const float* w2t, const float* b2,
                      float* gate, void* stream);

/* Same excitation with bf16 weights (throughput mode): w1p bf16 [C][ceil8(R)] (fc1.weight^T zero padded),
 * w2p bf16 [R][C] (fc2.weight^T) = tdeed_amd.engine.pack_se_bf16. */
int tdeed_se_gate_bf16_fwd(const float* pooled, int n_parts, float inv_cnt, int N, int C, int R,
                           const void* w1p, const float* b1, const void* w2p, const float* b2,
                           float* gate, void* stream);

/* ---- Gate-Shift(-Fuse) (model/impl/gsf.py:38-93, model/impl/gsm.py:89-116, eval BN) ------------
 * x: [B*T][h][w][C] (first F channels are gated).  Three launches:
 *  gate:   BN3d+ReLU+Conv3d(3x3x3, groups 2)+tanh -> gate fp32 [B*T][h][w][2]; also
 *          ysum[B*T][F] = sum_hw gate*x, xsum[B*T][F] = sum_hw x (fp32).  Two kernels inside: per-frame
 *          partial sums Q (each frame read once), then the temporal combine + tanh + spatial sums.
 *  weight: GSF only: fusion weight fw[B][F][T] = sigmoid(Conv2d(2->1,3x3) over the (c,t) plane
 *          of [mean shifted y ; mean r]) (gsf.py:61-78).
 *  apply:  out[B*T][h][w][Fp] : channels [0,F) = interleave(shift(y)*fw + r*(1-fw)) (GSM: fw==1
 *          i.e. shift(y)+r), channels [F,Fp) copied from x (Fp = F rounded up to 8). */
int tdeed_gsf_gate_fwd(const void* x, int B, int T, int h, int w, int C, int F,
                       const float* bn_scale, const float* bn_shift,
                       const float* wq /*[27][F] tap-major conv3D weight (VALU path)*/,
                       const void* wqf /*bf16 MFMA fragments, engine.pack_gsf_q_frags; NULL => VALU*/,
                       const float* b3d /*[2]*/,
                       float* Q /*scratch fp32 [B*T][h][w][6]*/, float* gate, float* ysum, float* xsum,
                       int dtype, void* stream);
/* The second launch of tdeed_gsf_gate_fwd alone (gates + spatial sums), for a site whose tap maps Q were made by the launch in
 * front of it (tdeed_bneck_gs_fwd's Q).  bf16. */
int tdeed_gsf_gate_sums_fwd(const void* x, int B, int T, int h, int w, int C, int F, const float* b3d, const float* Q,
                            float* gate, float* ysum, float* xsum, void* stream);
int tdeed_gsf_weight_fwd(const float* ysum, const float* xsum, int B, int T, int F, int hw,
                         const float* cw1 /*[2][3][3]*/, const float* cb1, const float* cw2,
                         const float* cb2, float* fw /*[B][F][T]*/, void* stream);
int tdeed_gsf_apply_fwd(const void* x, const float* gate, const float* fw /*NULL => GSM*/, int B,
                        int T, int h, int w, int C, int F, int Fp, void* out, int dtype,
                        void* stream);

/* weight + apply in one launch (GSF): every frame's block evaluates its own fusion weights from ysum/xsum first. */
int tdeed_gsf_apply_fused_fwd(const void* x, const float* gate, const float* ysum, const float* xsum,
                              const float* cw1, const float* cb1, const float* cw2, const float* cb2, int B,
                              int T, int h, int w, int C, int F, int Fp, void* out, int dtype, void* stream);
/* the same launch with the output left in SOURCE channel order (impl/gsf.py:79-91 without the final `c = i*(F/4)+j -> 2j+i`
 * interleave; bf16 only): for a caller that folds the interleave into the columns of the 1x1 conv reading the slice
 * (engine.gs_source_order_columns); everything stays in registers, one memory round trip per frame. */
int tdeed_gsf_blend_src_fwd(const void* x, const float* gate, const float* ysum, const float* xsum,
                            const float* cw1, const float* cb1, const float* cw2, const float* cb2, int B,
                            int T, int h, int w, int C, int F, int Fp, void* out, int dtype, void* stream);

/* ---- global average pool + positional encoding (model.py:133-137) --------------------------
 * x: [B*T][hw][C] -> feat [B][T][C] = mean_hw(x) + temp_enc[t][c] (temp_enc fp32 [T][C]).  rowstat (optional, fp32
 * [B*T][2]): LayerNorm mean / rstd over C of every stored feature row, for tdeed_sgp_front_fwd of the first SGP block. */
int tdeed_avgpool_posenc_fwd(const void* x, int B, int T, int hw, int C, const float* temp_enc,
                             void* feat, float* rowstat, int dtype, int dtype_out /* feat: dtype, or TDEED_F32 */, void* stream);

/* ---- SGP pyramid pieces (model/modules.py:58-363), NTC layout ------------------------------- */
/* channel LayerNorm of every row (modules.py:320-363): y[r][c] = (x-mu)/sqrt(var+eps)*w[c]+b[c];
 * rows = B*T, row strides ldx / ldy (elements) so the result can land in a slab of the mixer's
 * concat buffer. */
int tdeed_layernorm_fwd(const void* x, long ldx, int rows, int C, const float* w, const float* b,
                        float eps, void* y, long ldy, int dtype, void* stream);
/* SGPBlock front half (modules.py:161-184): given o = LN(x):
 *   y = x + fc(o)*relu(gfc(mean_T o)) + (convw(o)+convkw(o))*psi(o) + o
 * dw: fp32 packed per channel [C][2*ks+up+2]: psi[ks], convw[ks], convkw[up], fc, global_fc;
 * db: fp32 [5][C] biases in the order psi, convw, convkw, fc, global_fc. */
int tdeed_sgp_branch_fwd(const void* o, const void* x, int B, int T, int C, int ks, int up,
                         const float* dw, const float* db, void* y, int dtype, void* stream);
/* SGPMixer front half (modules.py:286-308): zn = LN1(z) already sits in slab 4 of cat, xn = LN2(x)
 * at T_lo.  Upsamples xn (linear, align_corners) and fills slabs 0-3 and 5 of
 * cat [B][T_hi][6C] = (out1,out2,out3,out4,zn,xu).  dw1/db1, dw2/db2 as in sgp_branch. */
int tdeed_mixer_branch_fwd(const void* xn, int B, int T_hi, int T_lo, int C, int ks, int up,
                           const float* dw1, const float* db1, const float* dw2,
                           const float* db2, void* cat, int dtype, void* stream);
/* ---- fused SGP launches (sgp_fused.hip): a block = sgp_front + two contractions, a mixer = mixer_front + three.
 * sgp_front:   y = x + LN(x) + fc*phi + (convw+convkw)*psi   (modules.py:159-184; LayerNorm 320-363 computed in-kernel)
 * mixer_front: cat [B][T_hi][6C] = (out1,out2,out3,out4,LN1(z),up(LN2(x_lo)))   (modules.py:286-308)
 * (out = y + mlp(GroupNorm16(y)) and concat_fc: tdeed_sgp_gemm_* below, csrc/sgp_gemm.hip) */
/* chsum (optional, fp32 [B][C][2]): per clip and channel the sum and sum of squares over T of the stored y, which
 * tdeed_sgp_gemm_gn_gelu takes for its GroupNorm statistics */
/* rowstat* (optional, fp32 [rows][2] = LayerNorm mean, rstd of every input row, as tdeed_avgpool_posenc_fwd leaves them):
 * taken instead of re-deriving the statistics from the clip's (T x C) slab */
/* rowstat*_parts: 0 = the (mean, rstd) form above; n > 0 = [n][rows][2] partial (sum, sum of squares) of each row over the n
 * column tiles of the tdeed_sgp_gemm_residual launch that stored the rows (summed in order; mean / rstd derived here).
 * dtype_cat (mixer): the six slabs may be stored as bf16 while z / x_lo are fp32 (fp32 residual stream, bf16 contraction). */
/* diagnostic: int64 [workgroups][16] phase time stamps of tdeed_sgp_front_fwd (wall_clock64, 10 ns ticks; slots 0..5 = start,
 * loads issued + row statistics, tile in LDS, LayerNorm applied, branches done, stored); null switches it off */
int tdeed_sgp_front_set_debug(void* buf);
int tdeed_sgp_front_fwd(const void* x, int B, int T, int C, int ks, int up, const float* ln_w, const float* ln_b, float eps,
                        const float* dw, const float* db, void* y, float* chsum, const float* rowstat, int rowstat_parts,
                        void* y16 /* optional, dtype fp32 only: a bf16 copy of y (the fc1 operand of wide models) */,
                        int dtype, void* stream);
int tdeed_mixer_front_fwd(const void* z, const void* xlo, int B, int T_hi, int T_lo, int C, int ks, int up,
                          const float* ln1_w, const float* ln1_b, const float* ln2_w, const float* ln2_b, float eps,
                          const float* dw1, const float* db1, const float* dw2, const float* db2, void* cat,
                          const float* rowstat_z, int rowstat_z_parts, const float* rowstat_x, int rowstat_x_parts,
                          int dtype, int dtype_cat, void* stream);
/* nn.GroupNorm(G, C) over (C/G x T) per clip (modules.py:115,186): x,y [B][T][C]. */
int tdeed_groupnorm_fwd(const void* x, int B, int T, int C, int G, const float* w, const float* b,
                        float eps, void* y, int dtype, void* stream);
/* nn.AdaptiveMaxPool1d over T (modules.py:64,76): [B][T_in][C] -> [B][T_out][C]. */
int tdeed_maxpool_fwd(const void* x, int B, int T_in, int T_out, int C, void* y, int dtype,
                      void* stream);
/* the same pooling with the LayerNorm statistics (mean, rstd over C, eps inside the sqrt; modules.py:353-357) of every pooled
 * row in rowstat [B*T_out][2] -- what tdeed_sgp_front_fwd of the next block takes instead of re-deriving them */
int tdeed_maxpool_rowstat_fwd(const void* x, int B, int T_in, int T_out, int C, void* y, float* rowstat, float eps, int dtype,
                              void* stream);

/* ---- heads (modules.py:366-387, model.py:141-146), eval (no dropout) ------------------------
 * x [rows][C]; w fp32 [n_out][C], b [n_out]; out fp32 [rows][n_out] (class logits and the
 * displacement column are rows of the same matrix: one read of x). */
int tdeed_heads_fwd(const void* x, int rows, int C, const float* w, const float* b, int n_out,
                    float* out, int dtype, void* stream);

/* ---- loss (model.py:208-211, 308-319) -------------------------------------------------------
 * Weighted CE (hard int64 labels, or soft [rows][K1] when soft!=NULL) + MSE(displ, labelD).
 * out fp32 [3]: total, ce, mse.  logits fp32 [rows][ld] (first K1 columns), displ = column
 * `displ_col` of the same matrix (or <0 for none). */
int tdeed_loss_fwd(const float* logits, int rows, int ld, int K1, const int64_t* hard,
                   const float* soft, const float* cls_w, int displ_col, const float* labelD,
                   float* out, void* stream);

/* ---- training path, first pieces (a13 backward, a14): see train.hip ------------------------------------------
 * loss_bwd: dhead = grad_scale * d(CE + MSE)/d(head_out), same arguments as tdeed_loss_fwd; columns that do not
 *           take part in the loss get 0.
 * heads_bwd: FCLayers backward (modules.py:366-376): dx (activation dtype, may be NULL), dw fp32 [n_out][C],
 *           db fp32 [n_out]; workspace of tdeed_heads_bwd_workspace() bytes (deterministic two-stage reduction).
 * adamw_step: torch.optim.AdamW update (modules.py:37-39, defaults betas (0.9,0.999) eps 1e-8 wd 0.01) fused over
 *           one flat fp32 buffer; `step` counts from 1; grad_scale pre-multiplies the gradient (1/world, 1/acc...). */
int tdeed_loss_bwd(const float* head_out, int rows, int ld, int K1, const int64_t* hard, const float* soft,
                   const float* cls_w, int displ_col, const float* labelD, float grad_scale, float* dhead,
                   void* stream);
/* joint-dataset double head (model.py:278-306): per-clip CE on the clip's own class slice (K1a | K1b columns, dataset[i] in
 * {1,2}, labels of dataset 2 shifted by K1a), mean over clips, + displacement MSE.  out [3] and/or dhead (either NULL).
 * hard: int64 [B*T], or soft: fp32 [B*T][K1a+K1b] label distributions (mixup with the double head: the 3-D label branch
 * of model.py:286-300).  Labels outside [0, K) of their slice are never used as an index: the loss comes back NaN
 * (the same holds for tdeed_loss_fwd / tdeed_loss_bwd; torch's cross_entropy raises in that case). */
int tdeed_loss2(const float* head_out, int B, int T, int ld, int K1a, int K1b, const int64_t* dataset, const int64_t* hard,
                const float* soft, const float* cls_w, int displ_col, const float* labelD, float grad_scale, float* out,
                float* dhead, void* stream);
long tdeed_heads_bwd_workspace(int rows, int C, int n_out);
int tdeed_heads_bwd(const float* dout, const void* x, int rows, int C, const float* w, int n_out, void* dx,
                    float* dw, float* db, void* workspace, int dtype, void* stream);
int tdeed_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                     void* stream);

/* ---- post-proc (modules.py:406-426): softmax over the first K1 columns then scatter-max of
 * frame t onto clamp(t - rne(displ), 0, T-1).  scores fp32 [B][T][K1] (zero-initialised inside),
 * cls int64 [B][T] = argmax. */
int tdeed_process_prediction(const float* head_out, int B, int T, int ld, int K1, int displ_col,
                             float* scores, int64_t* cls, void* stream);

/* ---- utility ------------------------------------------------------------------------------- */
int tdeed_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream);
/* the large dense weights' kernel copies in one launch: tab = device array of n entries of six 8-byte fields {const float* src
 * (an fp32 [R][C] matrix), long R, long C, void* dst ([R][C] in `dtype`, or NULL), void* dstT ([C][R] in `dtype`, or NULL),
 * long first_tile}; tiles = sum over the entries of ceil(R/32) * ceil(C/32), first_tile its running prefix */
int tdeed_multi_cast_transpose(const void* tab, int n, long tiles, int dtype, void* stream);
/* out[i] = idx[i] ? src[idx[i] - 1] : 0 for i < n (n a multiple of 8), out fp32 or bf16: all kernel-layout copies of the
 * master parameters (casts, transposes, MFMA fragment orders, zero pads) refreshed in one launch from a recorded index
 * table; replaces the per-tensor `.to(bf16)` / `.t().contiguous()` of a framework step. */
int tdeed_gather_cast(const float* src, const int* idx, long n, void* out, int dtype, void* stream);
int tdeed_fill_u8_hash(uint8_t* dst, long n, uint64_t seed, void* stream); /* synthetic clips */

/* ---- backward of the SGP encoder-decoder (training path; sgp_bwd.hip) ------------------------------------------
 * Activations and activation gradients share the forward dtype; parameter gradients are fp32.  Every parameter
 * gradient is produced as caller-owned per-workgroup partials (`part*`) folded in a fixed order: no float atomics,
 * a step is bit-reproducible. */
/* out[j] (+)= sum_p part[p][j], p < P, j < n */
int tdeed_reduce_partials(const float* part, int P, long n, float* out, int accumulate, void* stream);
/* the same over rows that are `stride` floats apart (several parameter groups side by side in one partial row) */
int tdeed_reduce_strided(const float* part, int P, long stride, long n, float* out, void* stream);
/* gradient write-out of a whole step in one launch: tab = device array of nt records {const float* src; long dst_offset;
 * long n; long first_chunk} (first_chunk = running sum of ceil(n / 4096) over the records before it), n_chunks = that sum
 * over all records; dst[off + i] = (accumulate ? dst[off + i] : 0) + scale * src[i]. */
int tdeed_multi_copy(const void* tab, int nt, long n_chunks, float* dst, float scale, int accumulate, void* stream);
/* the same write-out with the fold of per-workgroup partials inside.  tab: device array of nt entries of 8 int64
 * {src pointer, dst offset, n, first workgroup, P | cw << 32, pstride, cols, ld}: P == 1 copies n elements from a source of
 * `cols` contiguous elements per row, rows `ld` apart (4096 elements per workgroup); P > 1 writes
 * dst[off + j] = scale * sum_p src[p * pstride + j] with cw = tdeed_multi_fold_cw(P, n) columns per workgroup (the partials
 * tdeed_wgrad leaves when called with accumulate = -1).  n_wgs = total workgroups of all entries. */
int tdeed_multi_fold_cw(int P, long n);
int tdeed_multi_fold(const void* tab, int nt, long n_wgs, float* dst, float scale, int accumulate, void* stream);
/* mode 0: y = gelu(x); 1: y = dy * gelu'(x); 2: y = x + dy; 3: y = x * dy.  n elements, multiple of 8 */
int tdeed_eltwise(const void* x, const void* dy, void* y, long n, int mode, int dtype, void* stream);
/* [R][Cc] -> [Cc][R] (weight transposes for the input-gradient contractions) */
int tdeed_transpose(const void* x, int R, int Cc, void* y, int dtype, void* stream);
/* weight / bias gradient of a Conv1d(k=1) / 1x1 conv: dW[n][k] (+)= sum_m dY[m][n] X[m][k], db[n] (+)= sum_m dY[m][n]
 * (db may be NULL).  part_w fp32 [Z][N][K], part_b fp32 [Z][N], Z = tdeed_wgrad_slices(M, N, K).
 * accumulate = -1: the partials are left unfolded (dW / db may be NULL; part_b is filled when given): tdeed_multi_fold folds
 * them together with the gradient write-out. */
int tdeed_wgrad_slices(int M, int N, int K);
/* X0 (optional, bf16, M >= 4096): columns k < k0 of the X operand come from X0 (row stride ldx0): the gate-shift splice
 * [G | x[:, k0:]] of a conv1 operand is never materialised (as A0 / k0 of tdeed_gemm_fwd) */
int tdeed_wgrad(const void* dY, long ldy, const void* X, long ldx, const void* X0, long ldx0, int k0, int M, int N, int K,
                float* part_w, float* part_b, float* dW, float* db, int accumulate, int dtype, void* stream);
/* channel LayerNorm backward (modules.py:320-363): dx (+)= d/dx, dw/db [C].  part fp32 [tdeed_layernorm_bwd_blocks(rows)][2][C] */
int tdeed_layernorm_bwd_blocks(int rows);
int tdeed_layernorm_bwd(const void* x, long ldx, const void* dy, long ldy, int rows, int C, const float* w, float eps,
                        void* dx, int accumulate, float* part, float* dw, float* db, int dtype, void* stream);
/* GroupNorm(G) backward over NTC slabs.  part fp32 [B][2][C] */
int tdeed_groupnorm_bwd(const void* x, const void* dy, int B, int T, int C, int G, const float* w, float eps, void* dx,
                        int accumulate, float* part, float* dw, float* db, int dtype, void* stream);
/* depthwise-branch backward of SGPBlock (modules.py:164-170) and of either input of SGPMixer (modules.py:290-305):
 * o = branch input (row stride ldo); g_conv / g_inst / g_id = gradients of (convw+convkw)*psi, of fc*phi and of the
 * identity term (row stride ldg; SGPBlock passes one tensor three times, SGPMixer three slabs of d cat); d_o (row
 * stride ld_do); d_dw [C][2ks+up+2], d_db [5][C] in tdeed_sgp_branch_fwd's packed layouts.
 * part_w fp32 [B][C][2ks+up+2], part_b fp32 [B][5][C]. */
int tdeed_sgp_branch_bwd(const void* o, long ldo, const void* g_conv, const void* g_inst, const void* g_id, long ldg,
                         int B, int T, int C, int ks, int up, const float* dw, const float* db, void* d_o, long ld_do,
                         float* part_w, float* part_b, float* d_dw, float* d_db, int dtype, void* stream);
/* nn.Upsample(linear, align_corners=True) backward: d_xu [B][T_hi][.] (row stride ld) -> d_xn [B][T_lo][C] */
int tdeed_upsample_bwd(const void* d_xu, long ld, int B, int T_hi, int T_lo, int C, void* d_xn, int dtype, void* stream);
/* nn.AdaptiveMaxPool1d backward (first maximum of a window takes the gradient, like torch) */
int tdeed_maxpool_bwd(const void* x, const void* dy, int B, int T_in, int T_out, int C, void* dx, int dtype,
                      void* stream);


/* ---- train-mode trunk pieces (trunk_bwd.hip): BatchNorm with batch statistics, SE with kept intermediates, grouped
 * 3x3 backward.  Maps are channels-last [M][C] in the activation dtype; statistics and parameter gradients fp32. */
int tdeed_bn_slabs(long M);
/* statistics of the raw conv map z: mean/rstd (kept for the backward), a = w*rstd, b = bias - mean*a (for
 * tdeed_bn_apply), running_mean/var updated in place when non-NULL (momentum 0.1, unbiased variance, like
 * nn.BatchNorm2d).  part fp32 [tdeed_bn_slabs(M)][2][C] */
int tdeed_bn_train_stats(const void* z, long M, int C, const float* w, const float* bias, float eps, float momentum,
                         float* part, float* mean, float* rstd, float* a, float* b, float* run_mean, float* run_var,
                         int dtype, void* stream);
/* the finalisation alone, from partial sums a producer's epilogue wrote (see tdeed_gemm_fwd colpart, tdeed_gconv3x3_fwd
 * pooled_sq): P rows of per-channel sums (part_s) / sums of squares (part_q), pstride floats apart, covering M rows */
int tdeed_bn_finalize(const float* part_s, const float* part_q, long pstride, int P, long M, int C, const float* w,
                      const float* bias, float eps, float momentum, float* mean, float* rstd, float* a, float* b,
                      float* run_mean, float* run_var, void* stream);
/* out[s][j] = sum of part[p * pstride + j] over the rows p of slice s (slices of ceil(P / slices) rows), j < n; rows of out
 * are `ostride` floats apart.  A first fold for tdeed_bn_finalize when a producer left one partial row per (frame, band). */
int tdeed_fold_rows(const float* part, long pstride, int P, int n, int slices, float* out, long ostride, void* stream);
/* y = act(z * a[c] + b[c] + res) */
int tdeed_bn_apply(const void* z, long M, int C, const float* a, const float* b, const void* res, int relu, void* y,
                   int dtype, void* stream);
/* y = act(z * a + b + (res * ra + rb)): the same with the residual a raw conv output under its own BatchNorm affine (the
 * shortcut conv of a downsampling bottleneck: its normalised map is never written) */
int tdeed_bn_apply2(const void* z, long M, int C, const float* a, const float* b, const void* res, const float* ra,
                    const float* rb, int relu, void* y, int dtype, void* stream);
/* ... and y2 [M][Fp2]: a compact copy of output channels [0, F2) (zeros in [F2, Fp2)) -- the dense slice the next block's
 * gate-shift module reads (what tdeed_gsf_slice would make with a pass of its own); ra / rb may be NULL */
int tdeed_bn_apply_slice(const void* z, long M, int C, const float* a, const float* b, const void* res, const float* ra,
                         const float* rb, int relu, void* y, void* y2, int F2, int Fp2, int dtype, void* stream);
/* BatchNorm (training) backward; when relu != 0 the ReLU mask comes from y (the block's output) or, with y NULL (no
 * residual in front of the ReLU), from z through the forward affine fa, fb (y > 0 <=> fa*z + fb > 0: one map less to
 * read); d_res (optional) receives the masked gradient = gradient of the residual summed in before the ReLU.  sums fp32
 * [2][C]: sums[0] = db, sums[1] = dw (also copied to db / dw when those are non-NULL). */
int tdeed_bn_train_bwd(const void* z, const void* dy, const void* y, int relu, long M, int C, const float* mean,
                       const float* rstd, const float* w, const float* fa, const float* fb, float* part, float* sums,
                       void* dz, void* d_res, float* dw, float* db, int dtype, void* stream);
/* p[n][c] = mean_px x (x2 NULL: the SE squeeze) or sum_px x*x2 (gradient of the SE gate).  aff_on = 1 / 2: x / x2 is a
 * raw conv output and relu(in_a[c] * . + in_b[c]) (BatchNorm + ReLU) is applied on load; 0: none (in_a, in_b may be NULL) */
int tdeed_pool_rows(const void* x, const void* x2, int N, int hw, int C, const float* in_a, const float* in_b, int aff_on,
                    float* p, int dtype, void* stream);
/* SE excitation keeping the hidden units: w1t [C][R], w2t [R][C] */
int tdeed_se_train_fwd(const float* p, int N, int C, int R, const float* w1t, const float* b1, const float* w2t,
                       const float* b2, float* hid, float* gate, void* stream);
/* its backward: w1 [R][C], w2 [C][R]; d_pre2 [N][C], d_hid [N][R] feed tdeed_wgrad, d_p [N][C] is d(squeeze) */
int tdeed_se_train_bwd(const float* d_gate, const float* gate, const float* hid, int N, int C, int R, const float* w1,
                       const float* w2, float* d_pre2, float* d_hid, float* d_p, void* stream);
/* y[n][px][c] = x'[n][px][c] * s[n][c] + add[n][c] * add_scale (add may be NULL); x' = x, or with in_a / in_b given
 * relu(in_a[c] * x + in_b[c]) (x a raw conv output: BatchNorm + ReLU applied on load) */
int tdeed_scale_rows(const void* x, const float* s, const float* add, float add_scale, int N, int hw, int C,
                     const float* in_a, const float* in_b, void* y, int dtype, void* stream);
/* conv1's BatchNorm backward with the statistics from conv2's input-gradient launch (stride-1 bottlenecks, bf16):
 *   tdeed_gconv3x3_dgrad_stats: dx [N][Hi][Wi][C] = grouped 3x3 conv of dy with the flipped / transposed weights (wfrag_t,
 *     packed like tdeed_gconv3x3_fwd's wfrag; one / zero: fp32 [C] ones / zeros), and part_s / part_q fp32
 *     [N][tdeed_gconv3x3_parts(Hi, Wi, C, 1)][C]: per (frame, band) sums of g and g * (bz - bmean), g = dx * [bfa bz + bfb > 0]
 *     (bz: conv1's raw output, bfa / bfb its BatchNorm affine, bmean its batch mean);
 *   tdeed_bn_bwd_masked_from_parts: folds such partial rows (P rows, pstride floats apart) into sums fp32 [2][C]
 *     (d bias, d weight) and runs the apply pass dz = k1 g + k2 z + k3 with the ReLU mask recomputed from z. */
int tdeed_gconv3x3_dgrad_stats(const void* dy, int N, int Hi, int Wi, int C, int gw, const void* wfrag_t, const float* one,
                               const float* zero, void* dx, const void* bz, const float* bfa, const float* bfb,
                               const float* bmean, float* part_s, float* part_q, void* stream);
/* stride-2 blocks: tdeed_gconv3x3_bwd (stride 2, bf16) whose input-gradient launch leaves the same partial rows:
 * part_s / part_q fp32 [N * tdeed_gconv3x3_bwd_stats_bands(Hi)][C] */
int tdeed_gconv3x3_bwd_stats_bands(int Hi);
int tdeed_gconv3x3_bwd_stats_fits(int N, int Hi, int Wi, int C, int gw);
int tdeed_gconv3x3_bwd_stats(const void* x, const void* dy, int N, int Hi, int Wi, int C, int gw, const float* w,
                             const float* in_a, const float* in_b, void* dx, float* part, float* dw, const void* bz,
                             const float* bfa, const float* bfb, const float* bmean, float* part_s, float* part_q, void* stream);
int tdeed_bn_bwd_masked_from_parts(const void* z, const void* dy, long M, int C, const float* mean, const float* rstd,
                                   const float* w, const float* fa, const float* fb, const float* part_s, const float* part_q,
                                   long pstride, int P, float* sums, void* dz, int dtype, void* stream);
/* conv1 backward of a NARROW training bottleneck in one launch (csrc/trunk_bwd3.hip; bf16; (Co, Ci) = (64, 32), (128, 64),
 * (128, 128), (64, 64): RegNetY-800MF s1 / s2; with fa = 0, fb = 1 and no R / sink also conv3's backward there): from conv2's input gradient dY and conv1's raw output Z ([M][Co]) the BatchNorm + ReLU
 * backward dz1 (never stored), the input gradient dX [M][Ci] = dz1 @ W1 + shortcut gradient, masked by [X > 0] with the
 * gradient sink's column sums (as tdeed_gemm_dgrad), and the partial weight gradients dz1^T @ X.
 * fa / fb / mean / rstd / w: conv1's BatchNorm; sums fp32 [2][Co] = (sum g, sum g xhat) of its backward; Wt [Ci][Co] = W1^T;
 * R: shortcut gradient [M][Ci] (r_hi > 0: rows of the even pixels of an r_hi x r_wi frame) or NULL; bpart fp32
 * [grid][3][Ci] (or NULL), wpart fp32 [grid][Co][Ci], grid = tdeed_narrow_conv1_bwd_grid(M, Co, Ci).
 * Z == NULL (not at 128 <- 128): Z is X @ W1^T of exactly these operands and is recomputed in the launch instead of read. */
int tdeed_narrow_conv1_bwd_fits(int Co, int Ci);
int tdeed_narrow_conv1_bwd_grid(long M, int Co, int Ci);
int tdeed_narrow_conv1_bwd(const void* dY, const void* Z, long M, int Co, int Ci, const float* fa, const float* fb,
                           const float* mean, const float* rstd, const float* w, const float* sums, const void* X, const void* Wt,
                           const void* R, long ldr, int r_hi, int r_wi, void* dX, int use_mask, const void* bz,
                           const float* bmean, const void* bzd, const float* bmean_d, float* bpart, float* wpart, void* stream);
/* sums fp32 [2][C] = (sum_p part_s, rstd * sum_p part_q) over P partial rows pstride floats apart (what
 * tdeed_gconv3x3_dgrad_stats / tdeed_gconv3x3_bwd_stats leave) */
int tdeed_bn_sums_from_parts(const float* part_s, const float* part_q, long pstride, int P, int C, const float* rstd,
                             float* sums, void* stream);
/* SE + conv2-BatchNorm backward of a bottleneck without the d_y2 map (csrc/trunk_bwd2.hip; timm SEModule / BatchNorm2d under
 * autograd, /root/reference/model/model.py:265-324).  d = d(y2 * gate) [N][hw][C] (conv3's input gradient), z = conv2's raw
 * output, y2 = relu(fa * z + fb):
 *   tdeed_se_bn_bwd_sums:     sums fp32 [5][N][C] = per (frame, channel) sum over the pixels of
 *                             d*y2 | d*m | d*m*(z-mean) | m | m*(z-mean),  m = [fa z + fb > 0];  sums[0] is d(gate)
 *   tdeed_se_bn_bwd_finalize: out fp32 [2][C] = (sum g, sum g * xhat) with g = m * (d * gate + d_p / hw): the conv2
 *                             BatchNorm's (d bias, d weight), from the frame sums, gate [N][C] and d_p [N][C] = d(squeeze)
 *   tdeed_se_bn_bwd_apply:    dz [N][hw][C] = k1 * g + k2 * z + k3 (the BatchNorm input gradient), g formed on the fly */
int tdeed_se_bn_bwd_sums(const void* d, const void* z, int N, int hw, int C, const float* fa, const float* fb,
                         const float* mean, float* sums, int dtype, void* stream);
int tdeed_se_bn_bwd_finalize(const float* sums, const float* gate, const float* d_p, int N, int hw, int C, const float* rstd,
                             float* out, void* stream);
int tdeed_se_bn_bwd_apply(const void* d, const void* z, const float* gate, const float* d_p, int N, int hw, int C,
                          const float* fa, const float* fb, const float* mean, const float* rstd, const float* w,
                          const float* sums, void* dz, int dtype, void* stream);
/* "Gradient sink" (csrc/trunk_bwd2.hip, gemm.hip): the ReLU backward at a block's output and the statistics pass of the
 * BatchNorm backward behind it, applied by the kernels that PRODUCE the gradient (timm Bottleneck under autograd,
 * /root/reference/model/model.py:265-324).
 * tdeed_gemm_dgrad: C [M][N] = ((A [M][K] @ W [N][K]^T) + R) * [mask > 0]; with r_hi > 0 the residual R has rows only for the
 *   even pixels of an r_hi x r_wi frame (stride-2 shortcut); C2 (optional, [M][n2]) takes columns [0, n2) BEFORE the residual
 *   and C keeps only the residual there (gate-shift blocks); bpart (optional) fp32 [ceil(M/128)][3][N]: per-tile column sums of
 *   the stored v, v * (bz - bmean) and (bzd given) v * (bzd - bmean_d).
 * tdeed_gsf_add_cols_sink: dx[m][c] += (a + b)[m][c] * [mask > 0] for c < Fp, with the sums of what was ADDED in
 *   part fp32 [tdeed_gsf_add_cols_sink_parts(M, Fp, dtype)][3][Fp] (mask / part may be NULL).
 * tdeed_bn_bwd_from_parts: BatchNorm backward of z [M][C] for an already masked gradient g whose column sums lie in producer
 *   partials: partA PA rows of [3][C] (+ partB PB rows of [3][nB] for columns [0, nB)); q = 1 / 2 selects the product row;
 *   sums fp32 [2][C] = (d bias, d weight); dz (may be NULL) = the input gradient; tmp fp32 [2 * 64 * 3 * C]. */
int tdeed_gemm_dgrad(const void* A, long lda, int M, int K, int N, const void* W, long ldw, const void* R, long ldr, int r_hi,
                     int r_wi, void* C, long ldc, void* C2, long ldc2, int n2, const void* mask, long ldmask, const void* bz,
                     long ldbz, const float* bmean, const void* bzd, long ldbzd, const float* bmean_d, float* bpart, int dtype,
                     void* stream);
/* tdeed_gemm_dgrad for K = N = 320 over many rows on the register-stationary contraction (Wfrag = the packed [N][K] matrix,
 * engine.pack_ws_weights): residual and mask required, no stride-2 residual, no second statistics map;
 * bpart fp32 [tdeed_gemm_rs_grid(M)][3][N]. */
int tdeed_gemm_dgrad_rs(const void* A, long lda, int M, int K, int N, const void* Wfrag, const void* R, long ldr, void* C,
                        long ldc, void* C2, long ldc2, int n2, const void* mask, long ldmask, const void* bz, long ldbz,
                        const float* bmean, float* bpart, void* stream);
int tdeed_gsf_add_cols_sink_parts(long M, int Fp, int dtype);
int tdeed_gsf_add_cols_sink(const void* a, const void* b, long M, int C, int Fp, void* dx, const void* mask, long ldmask,
                            const void* bz, long ldbz, const float* bmean, const void* bzd, long ldbzd, const float* bmean_d,
                            float* part, int dtype, void* stream);
/* tdeed_gsf_add_cols_sink with the backward of the module's BatchNorm3d applied to b on load: b = the gradient at the
 * BatchNorm's output (d_bn of tdeed_gsf_bwd), bnx [M][Fp] its input (the dense slice), bn_sums fp32 [2][Fp] = (sum g,
 * sum g xhat) as tdeed_bn_bwd_from_parts leaves them, bn_mean / bn_rstd / bn_w [Fp]: batch statistics and weight
 * (torch.nn.BatchNorm3d under autograd, /root/reference/model/impl/gsf.py:38-93). */
int tdeed_gsf_add_cols_sink_bn(const void* a, const void* b, long M, int C, int Fp, void* dx, const void* mask, long ldmask,
                               const void* bz, long ldbz, const float* bmean, const void* bzd, long ldbzd,
                               const float* bmean_d, float* part, const void* bnx, const float* bn_sums, const float* bn_mean,
                               const float* bn_rstd, const float* bn_w, int dtype, void* stream);
int tdeed_bn_bwd_from_parts(const void* z, const void* g, long M, int C, const float* mean, const float* rstd, const float* w,
                            const float* partA, int PA, const float* partB, int PB, int nB, int q, float* tmp, float* sums,
                            void* dz, int dtype, void* stream);
/* grouped 3x3 backward: dx (may be NULL: a stride-1 input gradient is itself a grouped 3x3 conv of dy with the flipped,
 * transposed weights and can run on tdeed_gconv3x3_fwd's MFMA kernel) and dw (fp32, the forward's packed [G][9][gw][gw]).
 * part fp32 [tdeed_gconv_wgrad_slabs(N*Ho*Wo)][G*9*gw*gw] */
int tdeed_gconv_wgrad_slabs(long npix_out);
/* in_a, in_b (optional, bf16): x is the raw output of the conv in front, relu(in_a[c] * x + in_b[c]) is applied on load
 * (as in tdeed_gconv3x3_fwd) */
int tdeed_gconv3x3_bwd(const void* x, const void* dy, int N, int Hi, int Wi, int C, int gw, int stride, const float* w,
                       const float* in_a, const float* in_b, void* dx, float* part, float* dw, int dtype, void* stream);
/* mode 0: out[(f,yo,xo)] = in[(f,2yo,2xo)] (operand of the stride-2 shortcut conv); mode 1: out[(f,2yo,2xo)] += in[(f,yo,xo)] */
int tdeed_stride2_rows(const void* in, void* out, int F, int hi, int wi, int C, int mode, int dtype, void* stream);

/* avgpool + positional encoding backward: d x[f][p][c] = d feat[f][c]/hw, d temp_enc[t][c] = sum_b d feat[b][t][c] */
int tdeed_avgpool_posenc_bwd(const void* d_feat, int B, int T, int hw, int C, void* dx, float* d_temp_enc, int dtype,
                             void* stream);
/* stem weight gradient (the uint8 input needs none): dz [N][Ho][Wo][32] -> dw [32][3][3][3]; part fp32
 * [N * ceil(Ho/16)][864] */
int tdeed_stem_wgrad(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left, int crop_h, int crop_w,
                     int flip, const unsigned char* flip_mask /* NULL or [N], as tdeed_stem_fwd */, const void* dz,
                     float* part, float* dw, int dtype, void* stream);
/* tdeed_stem_wgrad with the stem BatchNorm's backward (torch.nn.BatchNorm2d of timm's stem ConvNormAct under autograd) applied
 * while the gradient rows are staged: g = the masked gradient at the BatchNorm's output, z = the raw stem output, sums fp32
 * [2][32] = (sum g, sum g xhat) as tdeed_bn_bwd_from_parts leaves them, mean / rstd / w [32].  bf16; geometry per
 * tdeed_stem_wgrad_bn_fits. */
int tdeed_stem_wgrad_bn_fits(int H, int W, int crop_h, int crop_w);
int tdeed_stem_wgrad_bn(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left, int crop_h,
                        int crop_w, int flip, const unsigned char* flip_mask, const void* g, const void* z, const float* sums,
                        const float* mean, const float* rstd, const float* w, float* part, float* dw, void* stream);

/* train-time augmentation inside Impl.forward (model.py:76-83, 154-157): per clip ColorJitter(hue), (saturation),
 * (brightness), (contrast), GaussianBlur(5) on the cropped frames (torchvision 0.18.1 float-tensor arithmetic restated in
 * csrc/augment.hip); RandomHorizontalFlip is the flip_mask of tdeed_stem_fwd.  prm fp32 [B][8] on the device =
 * {hue shift, saturation, brightness, contrast, blur sigma (0 = off), -, -, -}, identity = {0,1,1,1,0}.  frames uint8 (or
 * fp32 0..255) [B*T][3][H][W] -> out fp32 0..255 [B*T][3][crop_h][crop_w] (feed to tdeed_stem_fwd with frames_f32 = 1 and
 * no crop).  part: tdeed_augment_scratch_floats(B*T) floats; tmp: same size as out (blur staging). */
long tdeed_augment_scratch_floats(int N);
int tdeed_augment_clips(const void* frames, int frames_f32, int B, int T, int H, int W, int crop_top, int crop_left,
                        int crop_h, int crop_w, const float* prm, float* part, float* out, float* tmp, void* stream);

/* mixup (model.py:240-256): out[b] = lam[b]*a[b] + (1-lam[b])*b[b], uint8 clips -> fp32 frames for tdeed_stem_fwd(frames_f32=1) */
int tdeed_mix_frames(const uint8_t* a, const uint8_t* b, const float* lam, int B, long per_clip, float* out, void* stream);

/* ---- gate-shift-fuse backward (gsf_bwd.hip; forward tensors as saved by tdeed_gsf_gate_fwd / tdeed_gsf_weight_fwd).
 * tdeed_gsf_slice: dense copy xs [M][Fp] of the module's channels (cols >= F zero): the BatchNorm3d operand in training.
 * tdeed_gsf_bwd: from dA (gradient of the module output in conv1's operand layout, [N*hw][Fp]) to d_xs (direct part of
 *   d x, dense [M][Fp]) and d_bn (gradient at the BatchNorm3d output, ReLU mask applied, dense [M][Fp]); parameter
 *   gradients d_w3 [F][27] (conv3D.weight flattened), d_b3 [2], d_cw [2][18] (channel_conv1 | channel_conv2), d_cb [2].
 *   w3 = conv3D.weight as [F][27]; sa/sb = the BatchNorm3d affine of this step; scratch fp32
 *   [tdeed_gsf_bwd_scratch_floats(B,T,hw,F)].
 * tdeed_gsf_add_cols: dx[m][0:Fp] += a[m][:] + b[m][:] (dx row stride C). */
int tdeed_gsf_slice(const void* x, long M, int C, int F, int Fp, void* xs, int dtype, void* stream);
long tdeed_gsf_bwd_scratch_floats(int B, int T, int hw, int F);
/* tdeed_gsf_bwd_part_layout: where tdeed_gsf_bwd(d_w3 == NULL) leaves its parameter-gradient partials inside `scratch`
 *   (float offsets): out[6] = {off_cw, rows_cw, stride_cw, off_w3, rows_w3, stride_w3}; part_cw row columns 0..17
 *   channel_conv1 taps, 18 its bias, 19..36 channel_conv2 taps, 37 its bias; part_w3 row = conv3D.weight [F][27], 2 biases
 *   (replaces nothing in the reference: autograd accumulates these in model/impl/gsf.py:38-93's backward). */
int tdeed_gsf_bwd_part_layout(int B, int T, int hw, int F, long* out);
int tdeed_gsf_bwd(const void* x, const float* gate, const float* fw, const float* ysum, const float* xsum,
                  const void* dA, int B, int T, int h, int w, int C, int F, int Fp, const float* w3, const float* sa,
                  const float* sb, const float* cw1, const float* cw2, float* scratch, void* d_xs, void* d_bn,
                  float* d_w3, float* d_b3, float* d_cw, float* d_cb, int dtype, void* stream);
int tdeed_gsf_add_cols(const void* a, const void* b, long M, int C, int Fp, void* dx, int dtype, void* stream);
/* tdeed_gsf_bwd that also leaves the statistics of the module's BatchNorm3d backward: bn_part fp32
 * [tdeed_gsf_bwd_bn_parts(B,T,h,w,C,Fp)][3][Fp], rows 0 / 1 = per-workgroup sums of d_bn and d_bn * (x - bn_mean) (the layout
 * tdeed_bn_bwd_from_parts folds with q = 1); tdeed_gsf_bwd_bn_parts == 0: not served at this geometry.  In both entries
 * d_w3 == NULL leaves the parameter gradients as partials inside scratch (layout in gsf_bwd.hip) for the caller's fold. */
int tdeed_gsf_bwd_bn_parts(int B, int T, int h, int w, int C, int Fp);
int tdeed_gsf_bwd_stats(const void* x, const float* gate, const float* fw, const float* ysum, const float* xsum,
                        const void* dA, int B, int T, int h, int w, int C, int F, int Fp, const float* w3, const float* sa,
                        const float* sb, const float* cw1, const float* cw2, float* scratch, void* d_xs, void* d_bn,
                        float* d_w3, float* d_b3, float* d_cw, float* d_cb, const float* bn_mean, float* bn_part, int dtype,
                        void* stream);

/* ---- data-parallel gradient reduction over RCCL / xGMI (comm.hip) ---------------------------------------------------
 * New functionality (the reference is single-GPU: model/model.py:184-190): one process per GPU, one
 * communicator with its own high-priority HIP stream.  tdeed_comm_all_reduce enqueues an in-place SUM of buf[0,n) over all
 * ranks behind the work already given to `compute_stream` and returns at once (the backward goes on); tdeed_comm_join makes
 * `compute_stream` wait for every collective enqueued so far (call it before the optimizer reads the gradients; the 1/world
 * goes into tdeed_adamw_step's grad_scale).  No host synchronisation in either; legal under stream capture.  librccl is
 * dlopen'ed at the first tdeed_comm_* call.  id128: 128 opaque bytes made on rank 0 (tdeed_comm_unique_id) and handed to
 * every rank by the caller (torch.distributed broadcast, a file ...); tdeed_comm_init is collective over the ranks. */
int tdeed_comm_unique_id(void* id128);
int tdeed_comm_init(void** comm_out, const void* id128, int world, int rank);
int tdeed_comm_info(void* comm, int* world, int* rank);
int tdeed_comm_all_reduce(void* comm, void* buf, long n, int dtype, void* compute_stream);
/* the same sum as reduce-scatter + all-gather over the first floor(n / world) * world elements plus a plain all-reduce of the
 * (< world) elements behind them: for the large bucket on point-to-point xGMI; any n is legal */
int tdeed_comm_all_reduce_rs_ag(void* comm, void* buf, long n, int dtype, void* compute_stream);
int tdeed_comm_join(void* comm, void* compute_stream);
int tdeed_comm_destroy(void* comm);

/* ---- HIP graph capture of a launch sequence (replaces eager op-by-op dispatch) ---------------
 * begin: hipStreamBeginCapture(stream); end: EndCapture + Instantiate -> handle; launch replays. */
int tdeed_graph_begin(void* stream);
int tdeed_graph_end(void* stream, void** graph_exec);
int tdeed_graph_launch(void* graph_exec, void* stream);
int tdeed_graph_destroy(void* graph_exec);

/* ---- dense contractions of the SGP encoder-decoder with their prologue / epilogue (csrc/sgp_gemm.hip; round 5).
 * Replaces, per SGPBlock / SGPMixer of /root/reference/model/modules.py: self.gn + self.mlp[0] + GELU (134-138, 186, 316),
 * self.mlp[2] + the residual add (186, 316) with the AdaptiveMaxPool1d of the encoder (64, 75-77) and the row sums the next
 * LayerNorm needs (320-363), concat_fc + GELU (245-246, 307-308) with the channel sums the next GroupNorm needs.
 * Weights: MFMA fragments [N/16][tdeed_sgp_gemm_ksteps(K)][64][8] bf16 (zero padded).  Row tiles never straddle clips.
 *   form = 16 * MT + NT (tile = 16 MT rows x 64 NT features) from tdeed_sgp_gemm_form; tdeed_sgp_gemm_row_tiles(T, MT) row
 *   tiles per clip (NJ), tdeed_sgp_gemm_col_tiles(N, NT) column tiles (nct).
 * tdeed_sgp_gemm_gn_gelu:   H[B*T][N] bf16 = GELU(GroupNorm_G(y) . W^T + b); y [B*T][K] in dtype_a; chsum [parts][B][K][2]
 *   (sum, sum of squares over the clip's rows per channel, summed over `parts` in order).
 * tdeed_sgp_gemm_residual:  out = resid + H . W^T + b (H bf16 [B*T][K]; out, resid, pooled in dtype_o); rowstat_part
 *   [nct][B*T][2] = (sum, sum of squares) of each stored row over the column tile; pooled != NULL (needs T == 2 T_out):
 *   the max of each row pair [B*T_out][N] and rowstat_pool_part [nct][B*T_out][2].
 * tdeed_sgp_gemm_gelu_chsum: out = GELU(A . W^T + b) (A bf16 [B*T][K], out in dtype_o); chs_out [NJ][B][N][2] per-channel
 *   (sum, sum of squares) of the stored rows of each row tile. */
int tdeed_sgp_gemm_ksteps(int K);
int tdeed_sgp_gemm_row_tiles(int T, int MT);
int tdeed_sgp_gemm_col_tiles(int N, int NT);
int tdeed_sgp_gemm_form(int mode, int B, int T, int N, int K);
int tdeed_sgp_gemm_gn_gelu(const void* y, int B, int T, int K, const float* chsum, int chs_parts, const float* gn_w,
                           const float* gn_b, int G, float eps, const void* Wp, const float* bias, int N, void* H, int form,
                           int dtype_a, void* stream);
int tdeed_sgp_gemm_residual(const void* H, int B, int T, int K, const void* Wp, const float* bias, int N, const void* resid,
                            void* out, float* rowstat_part, void* pooled, float* rowstat_pool_part, int T_out, int form,
                            int dtype_o, void* stream);
int tdeed_sgp_gemm_gelu_chsum(const void* A, int B, int T, int K, const void* Wp, const float* bias, int N, void* out,
                              float* chs_out, void* out16 /* optional bf16 copy of out */, int form, int dtype_o,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif
