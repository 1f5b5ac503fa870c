// First pieces of the training path: the output end of the backward chain
// and the optimizer.  Loss backward (weighted CE hard/soft + MSE, reference model/model.py:308-319), heads
// backward (FCLayers, model/modules.py:366-376) and a fused multi-tensor AdamW over one flat fp32 buffer
// (BaseRGBModel.get_optimizer -> torch.optim.AdamW defaults, model/modules.py:37-39).
#include "common.h"

// ------------------------------------------------------------------------------------------ loss backward
// d(total)/d(head_out) for head_out [rows][ld] (columns [0,K1) logits, column displ_col the displacement).
//   hard labels: dlogit[r][k] = w[y_r] (softmax_rk - [k==y_r]) / sum_r w[y_r]
//   soft labels: dlogit[r][k] = (softmax_rk * sum_c w_c p_rc - w_k p_rk) / rows
//   MSE:         ddispl[r]    = 2 (d_r - labelD_r) / rows
// single block (rows = B*T is a few thousand); columns outside the loss get zero gradient.
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ head, int rows, int ld, int K1,
                                                       const int64_t* __restrict__ hard,
                                                       const float* __restrict__ soft,
                                                       const float* __restrict__ cls_w, int displ_col,
                                                       const float* __restrict__ labelD, float gscale,
                                                       float* __restrict__ dhead) {
  __shared__ float scratch[8];
  float den = 0.f;
  if (!soft)
    for (int r = threadIdx.x; r < rows; r += 256) {
      const int y = (int)hard[r];
      den += cls_w[(y >= 0 && y < K1) ? y : 0];                  // out-of-range labels are never indexed (tdeed_loss_fwd -> NaN)
    }
  den = soft ? (float)rows : block_sum<4>(den, scratch);
  const float inv = gscale / den;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float* lg = head + (long)r * ld;
    float* dg = dhead + (long)r * ld;
    float m = lg[0];
    for (int k = 1; k < K1; ++k) m = fmaxf(m, lg[k]);
    float s = 0.f;
    for (int k = 0; k < K1; ++k) s += expf(lg[k] - m);
    const float is = 1.0f / s;
    for (int k = K1; k < ld; ++k) dg[k] = 0.f;
    if (soft) {
      const float* p = soft + (long)r * K1;
      float wp = 0.f;
      for (int k = 0; k < K1; ++k) wp += cls_w[k] * p[k];
      for (int k = 0; k < K1; ++k) dg[k] = (expf(lg[k] - m) * is * wp - cls_w[k] * p[k]) * inv;
    } else {
      int y = (int)hard[r];
      if (y < 0 || y >= K1) y = 0;
      const float wy = cls_w[y];
      for (int k = 0; k < K1; ++k) dg[k] = wy * (expf(lg[k] - m) * is - (k == y ? 1.f : 0.f)) * inv;
    }
    if (displ_col >= 0 && labelD) dg[displ_col] = 2.0f * (lg[displ_col] - labelD[r]) * gscale / (float)rows;
  }
}

extern "C" int tdeed_loss_bwd(const float* head_out, int rows, int ld, int K1, const int64_t* hard, const float* soft,
                              const float* cls_w, int displ_col, const float* labelD, float grad_scale,
                              float* dhead, void* stream) {
  TD_CHECK(head_out && cls_w && dhead && (hard || soft), "loss_bwd: null pointer");
  TD_CHECK(rows > 0 && K1 > 0 && K1 <= ld && displ_col < ld, "loss_bwd: bad sizes");
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, head_out, rows, ld, K1, hard, soft,
                     cls_w, displ_col, labelD, grad_scale, dhead);
  TD_LAUNCH_CHECK("loss_bwd");
  return TDEED_OK;
}

// ------------------------------------------------------------------------------------------ heads backward
// out = x W^T + b  (x [rows][C] in T, W fp32 [n_out][C]):  dx = dout W (written in T),
// dW[o][c] = sum_r dout[r][o] x[r][c], db[o] = sum_r dout[r][o]   (fp32, deterministic two-stage reduction).
template <typename T>
__global__ __launch_bounds__(256) void heads_bwd_dx_kernel(const float* __restrict__ dout, int rows, int C,
                                                           const float* __restrict__ w, int n_out,
                                                           T* __restrict__ dx) {
  constexpr int EPC = Chunk<T>::N;
  const int cpr = C / EPC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows * cpr; i += (long)gridDim.x * 256) {
    const long r = i / cpr;
    const int c0 = (int)(i - r * cpr) * EPC;
    float a[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[e] = 0.f;
    for (int o = 0; o < n_out; ++o) {
      const float g = dout[r * n_out + o];
#pragma unroll
      for (int e = 0; e < EPC; ++e) a[e] = fmaf(g, w[(long)o * C + c0 + e], a[e]);
    }
    Chunk<T>::store(dx + r * C + c0, a);
  }
}

// grid (row slices, n_out): partial[s][o][c] over rows of slice s
template <typename T>
__global__ __launch_bounds__(256) void heads_bwd_dw_partial_kernel(const float* __restrict__ dout,
                                                                   const T* __restrict__ x, int rows, int C,
                                                                   int n_out, int rows_per, float* __restrict__ part,
                                                                   float* __restrict__ partb) {
  const int s = blockIdx.x, o = blockIdx.y;
  const int r0 = s * rows_per, r1 = min(rows, r0 + rows_per);
  __shared__ float scratch[8];
  float bsum = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a = fmaf(dout[(long)r * n_out + o], (float)x[(long)r * C + c], a);
    part[((long)s * n_out + o) * C + c] = a;
  }
  for (int r = r0 + threadIdx.x; r < r1; r += 256) bsum += dout[(long)r * n_out + o];
  bsum = block_sum<4>(bsum, scratch);
  if (threadIdx.x == 0) partb[(long)s * n_out + o] = bsum;
}

__global__ void heads_bwd_dw_reduce_kernel(const float* __restrict__ part, const float* __restrict__ partb, int S,
                                           int n_out, int C, float* __restrict__ dw, float* __restrict__ db) {
  const long n = (long)n_out * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n + n_out; i += (long)gridDim.x * blockDim.x) {
    float a = 0.f;
    if (i < n) {
      for (int s = 0; s < S; ++s) a += part[(long)s * n + i];
      dw[i] = a;
    } else {
      for (int s = 0; s < S; ++s) a += partb[(long)s * n_out + (i - n)];
      db[i - n] = a;
    }
  }
}

extern "C" long tdeed_heads_bwd_workspace(int rows, int C, int n_out) {
  const int S = rows >= 64 ? 64 : rows;
  return (long)S * n_out * (C + 1) * (long)sizeof(float);
}

extern "C" int tdeed_heads_bwd(const float* dout, const void* x, int rows, int C, const float* w, int n_out,
                               void* dx, float* dw, float* db, void* workspace, int dtype, void* stream) {
  TD_CHECK(dout && x && w && dw && db && workspace, "heads_bwd: null pointer");
  TD_CHECK(rows > 0 && C % 8 == 0 && n_out > 0, "heads_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int S = rows >= 64 ? 64 : rows;
  const int rows_per = (rows + S - 1) / S;
  float* part = (float*)workspace;
  float* partb = part + (long)S * n_out * C;
  if (dtype == TDEED_F32) {
    if (dx) hipLaunchKernelGGL(heads_bwd_dx_kernel<float>, dim3(cdiv((long)rows * (C / 4), 256)), dim3(256), 0, st, dout,
                               rows, C, w, n_out, (float*)dx);
    hipLaunchKernelGGL(heads_bwd_dw_partial_kernel<float>, dim3(S, n_out), dim3(256), 0, st, dout, (const float*)x, rows,
                       C, n_out, rows_per, part, partb);
  } else if (dtype == TDEED_BF16) {
    if (dx) hipLaunchKernelGGL(heads_bwd_dx_kernel<bf16_t>, dim3(cdiv((long)rows * (C / 8), 256)), dim3(256), 0, st, dout,
                               rows, C, w, n_out, (bf16_t*)dx);
    hipLaunchKernelGGL(heads_bwd_dw_partial_kernel<bf16_t>, dim3(S, n_out), dim3(256), 0, st, dout, (const bf16_t*)x,
                       rows, C, n_out, rows_per, part, partb);
  } else { tdeed_set_error("heads_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  hipLaunchKernelGGL(heads_bwd_dw_reduce_kernel, dim3(cdiv((long)n_out * C + n_out, 256)), dim3(256), 0, st, part, partb,
                     S, n_out, C, dw, db);
  TD_LAUNCH_CHECK("heads_bwd");
  return TDEED_OK;
}

// ------------------------------------------------------------------------------------------ fused AdamW
// torch.optim.AdamW semantics (decoupled decay, bias correction), all tensors fp32, one flat buffer per state:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// grad_scale multiplies g first (1/world for the data-parallel mean, 1/loss-scale ...).  16-byte vectorised.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                    float gscale) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * gscale;
      pv[e] *= 1.0f - lr * wd;
      mv[e] = b1 * mv[e] + (1.0f - b1) * gg;
      vv[e] = b2 * vv[e] + (1.0f - b2) * gg * gg;
      pv[e] -= (lr / bc1) * mv[e] / (sqrtf(vv[e]) / bc2s + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * gscale;
    float pv = p[i] * (1.0f - lr * wd);
    const float mv = b1 * m[i] + (1.0f - b1) * gg;
    const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
    pv -= (lr / bc1) * mv / (sqrtf(vv) / bc2s + eps);
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

extern "C" int tdeed_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                float grad_scale, void* stream) {
  TD_CHECK(param && grad && exp_avg && exp_avg_sq, "adamw: null pointer");
  TD_CHECK(n > 0 && step >= 1, "adamw: bad n/step");
  TD_CHECK((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
           "adamw: buffers must be 16-byte aligned");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  const long n4 = n >> 2;
  const int grid = (int)(n4 / 256 + 1 < 4096 ? n4 / 256 + 1 : 4096);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                     lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  TD_LAUNCH_CHECK("adamw");
  return TDEED_OK;
}

// =========================================================================== joint-dataset (double head) loss
// model.py:278-306: the class head is two heads side by side (K1a | K1b columns); clip i belongs to dataset ds[i] in
// {1, 2} and is scored on its own slice only: loss = sum_i CE_i / B with CE_i = sum_t w[y] nll / sum_t w[y] over the
// clip's T frames (labels of dataset 2 arrive shifted by K1a: update_labels_2heads), + the displacement MSE over all rows.
// One workgroup walks the clips (B is a few dozen, T a few hundred): fixed-order sums.
__global__ __launch_bounds__(256) void loss2_kernel(const float* __restrict__ head, int B, int T_len, int ld, int K1a,
                                                    int K1b, const int64_t* __restrict__ ds,
                                                    const int64_t* __restrict__ hard, const float* __restrict__ soft,
                                                    const float* __restrict__ cls_w,
                                                    int displ_col, const float* __restrict__ labelD, float gscale,
                                                    float* __restrict__ out, float* __restrict__ dhead) {
  __shared__ float scratch[8];
  float ce = 0.f, se_tot = 0.f, bad = 0.f;
  const long rows = (long)B * T_len;
  const int Ks = K1a + K1b;                                       // soft rows: distributions over both heads' columns
  for (int i = 0; i < B; ++i) {
    const bool first = ds[i] == 1;
    const int col0 = first ? 0 : K1a, K = first ? K1a : K1b;
    float num = 0.f, den = 0.f, se = 0.f;
    for (int t = threadIdx.x; t < T_len; t += 256) {
      const long r = (long)i * T_len + t;
      const float* lg = head + r * ld + col0;
      float m = lg[0];
      for (int k = 1; k < K; ++k) m = fmaxf(m, lg[k]);
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += expf(lg[k] - m);
      const float lse = m + logf(s);
      if (soft) {                                                 // mixup: -sum_c w_c p_c log softmax_c, mean over the T rows
        const float* p = soft + r * Ks + col0;
        for (int k = 0; k < K; ++k) num += cls_w[k] * p[k] * (lse - lg[k]);
      } else {
        int y = (int)hard[r] - col0;
        if (y < 0 || y >= K) { bad = 1.f; y = 0; }              // label outside the clip's own head: never indexed, loss -> NaN
        const float w = cls_w[y];
        num += w * (lse - lg[y]);
        den += w;
      }
      if (displ_col >= 0 && labelD) {
        const float d = head[r * ld + displ_col] - labelD[r];
        se += d * d;
      }
    }
    num = block_sum<4>(num, scratch);
    den = soft ? (float)T_len : block_sum<4>(den, scratch);
    se = block_sum<4>(se, scratch);
    ce += num / den / (float)B;
    se_tot += se;
    if (dhead) {
      const float inv = gscale / (den * (float)B);
      for (int t = threadIdx.x; t < T_len; t += 256) {
        const long r = (long)i * T_len + t;
        const float* lg = head + r * ld + col0;
        float* dg = dhead + r * ld;
        for (int k = 0; k < ld; ++k) dg[k] = 0.f;
        float m = lg[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, lg[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(lg[k] - m);
        const float is = 1.0f / s;
        if (soft) {
          const float* p = soft + r * Ks + col0;
          float wp = 0.f;
          for (int k = 0; k < K; ++k) wp += cls_w[k] * p[k];
          for (int k = 0; k < K; ++k) dg[col0 + k] = (expf(lg[k] - m) * is * wp - cls_w[k] * p[k]) * inv;
        } else {
          int y = (int)hard[r] - col0;
          if (y < 0 || y >= K) y = 0;
          const float wy = cls_w[y];
          for (int k = 0; k < K; ++k) dg[col0 + k] = wy * (expf(lg[k] - m) * is - (k == y ? 1.f : 0.f)) * inv;
        }
        if (displ_col >= 0 && labelD) dg[displ_col] = 2.0f * (head[r * ld + displ_col] - labelD[r]) * gscale / (float)rows;
      }
    }
  }
  bad = block_sum<4>(bad, scratch);
  if (threadIdx.x == 0 && out) {
    const float mse = (displ_col >= 0 && labelD) ? se_tot / (float)rows : 0.f;
    if (bad > 0.f) ce = __builtin_nanf("");                     // torch's cross_entropy raises here; a NaN loss is this path's alarm
    out[0] = ce + mse;
    out[1] = ce;
    out[2] = mse;
  }
}

// out fp32 [3] (total, CE, MSE) and/or dhead fp32 [B*T][ld] (either may be NULL); cls_w has max(K1a, K1b) entries.
// hard: int64 [B*T] labels over the concatenated heads, or soft: fp32 [B*T][K1a+K1b] (mixup; model.py:278-306 with 3-D labels).
// A hard label outside its clip's head slice is never used as an index: the row gets class 0 and the loss becomes NaN.
extern "C" int tdeed_loss2(const float* head_out, int B, int T, int ld, int K1a, int K1b, const int64_t* dataset,
                           const int64_t* hard, const float* soft, const float* cls_w, int displ_col, const float* labelD,
                           float grad_scale, float* out, float* dhead, void* stream) {
  TD_CHECK(head_out && dataset && (hard || soft) && cls_w && (out || dhead), "loss2: null pointer");
  TD_CHECK(B > 0 && T > 0 && K1a > 0 && K1b > 0 && K1a + K1b <= ld && displ_col < ld, "loss2: bad sizes");
  hipLaunchKernelGGL(loss2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, head_out, B, T, ld, K1a, K1b, dataset, hard,
                     soft, cls_w, displ_col, labelD, grad_scale, out, dhead);
  TD_LAUNCH_CHECK("loss2");
  return TDEED_OK;
}
