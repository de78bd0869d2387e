// slm_depth.hip -- "next" row f2: depth map -> per-frame target (reference depth_preprocessing,
// utils/data_loader.py:333-523).  All float32 steps follow the reference's operation order
// (back-projection = one rounded product + two fused multiply-adds, like torch.matmul on the 3x3
// intrinsics; central differences, cross product, F.normalize), so points are bit-exact against
// the reference and normals agree to float32 rounding.  HBM-bound stencil / scan / gather work:
//   k_dp_base / k_dp_dilate   invalid-pixel rules incl. the box dilations of torch_dilate
//   k_dp_points               final invalid map, back-projection, NaN marking
//   k_dp_normals              getN ("naive" central differences or colour-weighted 8 neighbours)
//   rocPRIM exclusive scan    index_map = running count of valid pixels (row-major)
//   k_dp_gather               compacted points / normals / colours / radii / confidences / semantics
//   k_dp_dist2edge            distance to the nearest class-boundary pixel (normalised image coords)
//   k_dp_warp / k_dp_ssim     stereo confidence (default CLI setting): the image warped through K * stereo_T at
//                             the back-projected depth (Project3D + grid_sample) and its 7x7-window SSIM against
//                             the image itself (skimage.metrics.structural_similarity semantics), mean over channels
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <string>

#include "slm_sem.h"

void slm_set_error_text(const char* msg);   // slm_api.hip

struct slm_depth {
  int H = 0, W = 0;
  uint8_t *m0 = nullptr, *m1 = nullptr;   // invalid-map ping-pong
  float* pcd = nullptr;                   // (H,W,3), NaN where invalid
  float* nrm = nullptr;                   // (H,W,3)
  int32_t *flag = nullptr, *idx = nullptr;   // valid flags (int) and their exclusive scan
  int32_t* total = nullptr;
  float* warp = nullptr;                  // (3,H,W) warped image (stereo confidence)
  float* ssim = nullptr;                  // (H,W) channel-mean SSIM
  void* tmp = nullptr;
  size_t cap_tmp = 0;
  SemScratch sem;
};

namespace {

#define DCHK(expr)                                                        \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) {                                               \
      slm_set_error_text((std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
      return SLM_ERR_HIP;                                                 \
    }                                                                     \
  } while (0)

int dfail(int code, const char* msg) {
  slm_set_error_text(msg);
  return code;
}

// superv1 base map: ~valid_mask | (seg in del classes)
__global__ void __launch_bounds__(256) k_dp_base(slm_depth_config c, slm_depth_inputs in, uint8_t* __restrict__ m) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.H * c.W) return;
  bool inval = in.valid_mask ? !in.valid_mask[p] : false;
  if (in.seg)
    for (int k = 0; k < c.n_del_classes; ++k) inval = inval || in.seg[p] == c.del_classes[k];
  m[p] = inval;
}

// torch_dilate: k x k box 'same' convolution > 0 (pads (k-1)/2 before, the rest after), with
// optional negation of the input and of the output (the superv1 rule is ~dilate(~x))
__global__ void __launch_bounds__(256) k_dp_dilate(int H, int W, int k, int neg_in, int neg_out,
                                                    const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p % W, lo = (k - 1) / 2;
  bool any = false;
  for (int dy = 0; dy < k && !any; ++dy) {
    const int yy = y - lo + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = 0; dx < k; ++dx) {
      const int xx = x - lo + dx;
      if (xx < 0 || xx >= W) continue;
      const bool v = src[yy * W + xx] != 0;
      if (neg_in ? !v : v) {
        any = true;
        break;
      }
    }
  }
  dst[p] = neg_out ? !any : any;
}

// final invalid map + back-projection (float32, the reference's operation order)
__global__ void __launch_bounds__(256) k_dp_points(slm_depth_config c, slm_depth_inputs in, uint8_t* __restrict__ m,
                                                    int have_base, float* __restrict__ pcd) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.H * c.W) return;
  const int y = p / c.W, x = p % c.W;
  const float d = in.depth[p];
  bool inval = have_base ? (m[p] != 0) : false;
  if (c.data_mode == 0) {
    if (c.raft_stereo && x < (int)(0.05 * c.W)) inval = true;
    inval = inval || d <= 0.0f || d > 1.5f;
  } else {
    if (c.load_depth) {
      inval = inval || d == 0.0f;
      // quirk kept: the reference slices ROWS with a width-derived bound (data_loader.py:411-412)
      if (y < (int)(0.1 * c.W)) inval = true;
    } else {
      if (y < (int)(c.depth_width_range[0] * c.W)) inval = true;
      if (y >= (int)(c.depth_width_range[1] * c.W)) inval = true;
    }
    if (in.seg)
      for (int k = 0; k < c.n_del_classes; ++k) inval = inval || in.seg[p] == c.del_classes[k];
  }
  m[p] = inval;
  const float u = (float)x, v = (float)y;
  const float nanv = __int_as_float(0x7fc00000);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float acc = c.inv_K[3 * i] * u;
    acc = __builtin_fmaf(c.inv_K[3 * i + 1], v, acc);
    acc = __builtin_fmaf(c.inv_K[3 * i + 2], 1.0f, acc);
    pcd[3 * p + i] = inval ? nanv : d * acc;
  }
}

// Project3D + F.grid_sample(bilinear, zeros, align_corners=False) of the image at the back-projected
// raw depth (depth/monodepth2/layers.py:141-192, data_loader.py:362-366), float32 like the reference
__global__ void __launch_bounds__(256) k_dp_warp(slm_depth_config c, slm_depth_inputs in, float* __restrict__ warp) {
#pragma clang fp contract(off)
  const int HW = c.H * c.W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int y = p / c.W, x = p % c.W;
  const float d = in.depth[p];
  const float u = (float)x, v = (float)y;
  float cam[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float acc = c.inv_K[3 * i] * u;
    acc = __builtin_fmaf(c.inv_K[3 * i + 1], v, acc);
    acc = __builtin_fmaf(c.inv_K[3 * i + 2], 1.0f, acc);
    cam[i] = d * acc;
  }
  float q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    q[i] = cam[0] * c.stereo_P[4 * i] + cam[1] * c.stereo_P[4 * i + 1] + cam[2] * c.stereo_P[4 * i + 2] + c.stereo_P[4 * i + 3];
  const float den = q[2] + 1e-7f;
  const float gx = (q[0] / den / (float)(c.W - 1) - 0.5f) * 2.0f;
  const float gy = (q[1] / den / (float)(c.H - 1) - 0.5f) * 2.0f;
  const float ix = ((gx + 1.0f) * (float)c.W - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * (float)c.H - 1.0f) / 2.0f;
  const float nanv = __int_as_float(0x7fc00000);
  if (!(isfinite(ix) && isfinite(iy))) {
    for (int k = 0; k < 3; ++k) warp[(size_t)k * HW + p] = nanv;
    return;
  }
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  // coordinates far outside the image: every tap is padding
  const bool far = fabsf(x0f) > 1e8f || fabsf(y0f) > 1e8f;
  const int x0 = far ? -2 : (int)x0f, y0 = far ? -2 : (int)y0f;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int xx = x0 + dx, yy = y0 + dy;
      if (xx < 0 || xx >= c.W || yy < 0 || yy >= c.H) continue;
      const float w = (dx ? wx1 : wx0) * (dy ? wy1 : wy0);
      for (int k = 0; k < 3; ++k) acc[k] += w * in.color[(size_t)k * HW + (size_t)yy * c.W + xx];
    }
  for (int k = 0; k < 3; ++k) warp[(size_t)k * HW + p] = acc[k];
}

// scipy.ndimage reflect: (d c b a | a b c d | d c b a)
__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i - 1;
  if (i >= n) i = 2 * n - 1 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// skimage.metrics.structural_similarity(warp, image, channel_axis=0, full=True).mean(0): 7x7 uniform window
// (separable, float32 after each pass like scipy.ndimage.uniform_filter), sample covariance, data_range 2
__global__ void __launch_bounds__(256) k_dp_ssim(int H, int W, const float* __restrict__ warp, const float* __restrict__ img,
                                                  float* __restrict__ out) {
#pragma clang fp contract(off)
  const int HW = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p % W;
  const float C1 = (float)((0.01 * 2.0) * (0.01 * 2.0)), C2 = (float)((0.03 * 2.0) * (0.03 * 2.0));
  const float cov_norm = (float)(49.0 / 48.0);
  float total = 0.0f;
  for (int ch = 0; ch < 3; ++ch) {
    const float* A = warp + (size_t)ch * HW;
    const float* B = img + (size_t)ch * HW;
    double s[5] = {0, 0, 0, 0, 0};          // horizontal pass over the float32 results of the vertical pass
    for (int dx = -3; dx <= 3; ++dx) {
      const int xx = reflect_idx(x + dx, W);
      double v[5] = {0, 0, 0, 0, 0};
      for (int dy = -3; dy <= 3; ++dy) {
        const int yy = reflect_idx(y + dy, H);
        const float a = A[(size_t)yy * W + xx], b = B[(size_t)yy * W + xx];
        v[0] += (double)a;
        v[1] += (double)b;
        v[2] += (double)(a * a);
        v[3] += (double)(b * b);
        v[4] += (double)(a * b);
      }
      for (int k = 0; k < 5; ++k) s[k] += (double)(float)(v[k] / 7.0);
    }
    const float ux = (float)(s[0] / 7.0), uy = (float)(s[1] / 7.0), uxx = (float)(s[2] / 7.0), uyy = (float)(s[3] / 7.0),
                uxy = (float)(s[4] / 7.0);
    const float vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
    const float A1 = 2.0f * ux * uy + C1, A2 = 2.0f * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
    total += (A1 * A2) / (B1 * B2);
  }
  out[p] = total / 3.0f;
}

struct f3 {
  float x, y, z;
};
__device__ __forceinline__ f3 ld3(const float* a, int H, int W, int y, int x) {
  const float nanv = __int_as_float(0x7fc00000);
  if (y < 0 || y >= H || x < 0 || x >= W) return {nanv, nanv, nanv};   // NaN padding
  const float* q = a + 3 * ((size_t)y * W + x);
  return {q[0], q[1], q[2]};
}
__device__ __forceinline__ f3 sub3(f3 a, f3 b) {
#pragma clang fp contract(off)
  return {a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ f3 add3(f3 a, f3 b) {
#pragma clang fp contract(off)
  return {a.x + b.x, a.y + b.y, a.z + b.z};
}
__device__ __forceinline__ f3 scl3(f3 a, float s) {
#pragma clang fp contract(off)
  return {a.x * s, a.y * s, a.z * s};
}
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
#pragma clang fp contract(off)
  const float x0 = a.y * b.z, x1 = a.z * b.y, y0 = a.z * b.x, y1 = a.x * b.z, z0 = a.x * b.y, z1 = a.y * b.x;
  return {x0 - x1, y0 - y1, z0 - z1};
}

// getN: normals from the NaN-padded vertex map; valid = no NaN in N and in the point
__global__ void __launch_bounds__(256) k_dp_normals(int H, int W, int model, const float* __restrict__ pcd,
                                                     const float* __restrict__ color, float* __restrict__ nrm,
                                                     int32_t* __restrict__ flag) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p % W;
  f3 N;
  if (model == 0) {
    const f3 hL = ld3(pcd, H, W, y, x - 1), hR = ld3(pcd, H, W, y, x + 1);
    const f3 hD = ld3(pcd, H, W, y - 1, x), hU = ld3(pcd, H, W, y + 1, x);   // names as in the reference
    N = cross3(sub3(hR, hL), sub3(hD, hU));
  } else {
    // neighbours in the reference's order: L, LU, U, RU, R, RD, D, DL
    const int oy[8] = {0, -1, -1, -1, 0, 1, 1, 1}, ox[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
    const f3 cen = ld3(pcd, H, W, y, x);
    const size_t HW = (size_t)H * W;
    f3 h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int yy = y + oy[k], xx = x + ox[k];
      float w;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
        w = __int_as_float(0x7fc00000);
      } else {
        const size_t q = (size_t)yy * W + xx;
        const float d0 = fabsf(color[q] - color[p]), d1 = fabsf(color[HW + q] - color[HW + p]),
                    d2 = fabsf(color[2 * HW + q] - color[2 * HW + p]);
        w = expf(-(((d0 + d1) + d2) / 3.0f));
      }
      h[k] = scl3(sub3(ld3(pcd, H, W, yy, xx), cen), w);
    }
    N = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 7; ++a) {
      f3 rest = h[a + 1];
#pragma unroll
      for (int b = a + 2; b < 8; ++b) rest = add3(rest, h[b]);
      N = add3(N, cross3(h[a], rest));
    }
  }
  // F.normalize(dim=-1): N / max(||N||, 1e-12)
  const float nn = sqrtf(((N.x * N.x) + (N.y * N.y)) + (N.z * N.z));
  const float den = fmaxf(nn, 1e-12f);
  const f3 U = {N.x / den, N.y / den, N.z / den};
  nrm[3 * p] = U.x;
  nrm[3 * p + 1] = U.y;
  nrm[3 * p + 2] = U.z;
  const bool ok = !(isnan(U.x) || isnan(U.y) || isnan(U.z)) &&
                  !(isnan(pcd[3 * p]) || isnan(pcd[3 * p + 1]) || isnan(pcd[3 * p + 2]));
  flag[p] = ok ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_dp_gather(slm_depth_config c, slm_depth_inputs in, slm_depth_outputs o, const float* __restrict__ ssim,
                                                    const float* __restrict__ pcd, const float* __restrict__ nrm,
                                                    const int32_t* __restrict__ flag, const int32_t* __restrict__ idx) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int HW = c.H * c.W;
  if (p >= HW) return;
  const bool ok = flag[p] != 0;
  if (o.valid) o.valid[p] = ok;
  if (o.index_map) o.index_map[p] = ok ? idx[p] : -1;
  if (!ok) return;
  const int t = idx[p];
  const int y = p / c.W, x = p % c.W;
  if (o.points)
    for (int k = 0; k < 3; ++k) o.points[3 * t + k] = pcd[3 * p + k];
  if (o.norms)
    for (int k = 0; k < 3; ++k) o.norms[3 * t + k] = nrm[3 * p + k];
  if (o.colors)
    for (int k = 0; k < 3; ++k) o.colors[3 * t + k] = in.color[(size_t)k * HW + p];
  if (o.radii) {
    // Z = -depth (data_loader.py:447); float32 depth / (sqrt(2) * fx * clamp(|n_z|, 0.26, 1)) in float64
    const double nz = fmin(fmax(fabs((double)nrm[3 * p + 2]), 0.26), 1.0);
    o.radii[t] = (double)(-in.depth[p]) / (sqrt(2.0) * (double)c.fx * nz);
  }
  if (o.confs) {
#pragma clang fp contract(off)
    const float su = (float)x / (float)c.W, sv = (float)y / (float)c.H;
    const float a = 2.0f * su - 1.0f, b = 2.0f * sv - 1.0f;
    const float dc2 = a * a + b * b;
    float conf = expf(-dc2 * (float)c.divterm);
    if (ssim) conf = 0.5f * conf + 0.5f * (1.0f / (1.0f + expf(-ssim[p])));   // data_loader.py:477-479
    o.confs[t] = conf;
  }
  if (in.seg && o.seg) o.seg[t] = in.seg[p];
  if (in.seg_conf && o.seg_conf) {
    const int C = c.num_classes;
    double v[SLM_MAX_CLASSES], mx = -1e300, den = 0.0;
    for (int k = 0; k < C; ++k) {
      v[k] = (double)in.seg_conf[(size_t)k * HW + p];
      mx = fmax(mx, v[k]);
    }
    for (int k = 0; k < C; ++k) {
      v[k] = exp(v[k] - mx);
      den += v[k];
    }
    for (int k = 0; k < C; ++k) o.seg_conf[(size_t)t * C + k] = v[k] / den;
  }
}

// dist2edge: nearest boundary pixel of the point's class, in image coordinates normalised by
// (W, H); boundary coordinates are float32 quotients like the reference's (data_loader.py:508)
__global__ void __launch_bounds__(256) k_dp_dist2edge(slm_depth_config c, int T, const float* __restrict__ points,
                                                       const int32_t* __restrict__ seg, const float2* __restrict__ edge_xy,
                                                       int e0, int e1, int e2, int e3, int e4, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int off[5] = {e0, e1, e2, e3, e4};
  const int cls = seg[t];
  double best = 0.0;
  if (cls >= 0 && cls < c.num_classes && off[cls + 1] > off[cls]) {
    const double X = (double)points[3 * t], Y = (double)points[3 * t + 1], Z = (double)points[3 * t + 2] + 1e-8;
    const double sx = (X * (double)c.fx / Z + (double)c.cx) / (double)c.W;
    const double sy = (Y * (double)c.fy / Z + (double)c.cy) / (double)c.H;
    double d2 = 1e300;
    for (int e = off[cls]; e < off[cls + 1]; ++e) {
      const float2 q = edge_xy[e];
      const double ex = (double)(q.x / (float)c.W), ey = (double)(q.y / (float)c.H);
      const double dx = sx - ex, dy = sy - ey;
      d2 = fmin(d2, dx * dx + dy * dy);
    }
    best = sqrt(d2);
  }
  out[t] = best;
}

template <typename T>
hipError_t dgrow(T*& p, size_t n) {
  if (p) return hipSuccess;
  return hipMalloc((void**)&p, n * sizeof(T));
}

}  // namespace

extern "C" {

int slm_depth_create(int32_t H, int32_t W, slm_depth** out) {
  if (!out || H < 8 || W < 8) return dfail(SLM_ERR_INVALID, "slm_depth_create: bad argument");
  if (slm_device_count() < 1) return dfail(SLM_ERR_NO_DEVICE, "slm_depth_create: no HIP device visible");
  slm_depth* d = new slm_depth();
  d->H = H;
  d->W = W;
  const size_t n = (size_t)H * W;
  hipError_t e = dgrow(d->m0, n);
  if (e == hipSuccess) e = dgrow(d->m1, n);
  if (e == hipSuccess) e = dgrow(d->pcd, 3 * n);
  if (e == hipSuccess) e = dgrow(d->nrm, 3 * n);
  if (e == hipSuccess) e = dgrow(d->flag, n);
  if (e == hipSuccess) e = dgrow(d->idx, n);
  if (e == hipSuccess) e = dgrow(d->total, 2);
  if (e == hipSuccess) e = dgrow(d->warp, 3 * n);
  if (e == hipSuccess) e = dgrow(d->ssim, n);
  if (e != hipSuccess) {
    slm_set_error_text((std::string("slm_depth_create: ") + hipGetErrorString(e)).c_str());
    slm_depth_destroy(d);
    return SLM_ERR_HIP;
  }
  *out = d;
  return SLM_OK;
}

int slm_depth_destroy(slm_depth* d) {
  if (!d) return SLM_OK;
  void* ptrs[] = {d->m0, d->m1, d->pcd, d->nrm, d->flag, d->idx, d->total, d->tmp, d->warp, d->ssim};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  sem_free(d->sem);
  delete d;
  return SLM_OK;
}

int slm_depth_preprocess(slm_depth* d, const slm_depth_config* cfg, const slm_depth_inputs* in,
                         const slm_depth_outputs* out, int32_t* n_valid_host, void* stream) {
  if (!d || !cfg || !in || !out) return dfail(SLM_ERR_INVALID, "slm_depth_preprocess: null argument");
  if (cfg->H != d->H || cfg->W != d->W) return dfail(SLM_ERR_INVALID, "slm_depth_preprocess: image size differs from slm_depth_create");
  if (!in->depth || !in->color) return dfail(SLM_ERR_INVALID, "slm_depth_preprocess: null device pointer");
  if (cfg->n_del_classes < 0 || cfg->n_del_classes > 3 || (cfg->n_del_classes > 0 && !in->seg))
    return dfail(SLM_ERR_INVALID, "slm_depth_preprocess: del_seg_classes needs the segmentation image");
  if (in->seg && (cfg->num_classes < 1 || cfg->num_classes > SLM_MAX_CLASSES))
    return dfail(SLM_ERR_UNSUPPORTED, "slm_depth_preprocess: num_classes must be 1..4");
  hipStream_t st = (hipStream_t)stream;
  const int H = d->H, W = d->W, HW = H * W;
  const dim3 grid((HW + 255) / 256), blk(256);
  // 0. stereo confidence from the raw depth (before anything is invalidated)
  const float* ssim = nullptr;
  if (cfg->use_ssim_conf) {
    hipLaunchKernelGGL(k_dp_warp, grid, blk, 0, st, *cfg, *in, d->warp);
    hipLaunchKernelGGL(k_dp_ssim, grid, blk, 0, st, H, W, d->warp, in->color, d->ssim);
    ssim = d->ssim;
    if (out->disp_conf) DCHK(hipMemcpyAsync(out->disp_conf, d->ssim, sizeof(float) * (size_t)HW, hipMemcpyDeviceToDevice, st));
  }
  // 1. invalid map
  int have_base = 0;
  uint8_t* m = d->m0;
  if (cfg->data_mode == 0) {
    hipLaunchKernelGGL(k_dp_base, grid, blk, 0, st, *cfg, *in, d->m0);
    have_base = 1;
    const int k = cfg->dilate_invalid_kernel;
    if (cfg->raft_stereo) {
      if (k > 0) {
        hipLaunchKernelGGL(k_dp_dilate, grid, blk, 0, st, H, W, k, 0, 0, d->m0, d->m1);
        m = d->m1;
      }
    } else if (k > 0) {
      hipLaunchKernelGGL(k_dp_dilate, grid, blk, 0, st, H, W, k, 1, 1, d->m0, d->m1);       // ~dilate(~x, k)
      hipLaunchKernelGGL(k_dp_dilate, grid, blk, 0, st, H, W, 2 * k, 0, 0, d->m1, d->m0);   // dilate(., 2k)
      m = d->m0;
    }
  }
  hipLaunchKernelGGL(k_dp_points, grid, blk, 0, st, *cfg, *in, m, have_base, d->pcd);
  if (out->inval) DCHK(hipMemcpyAsync(out->inval, m, (size_t)HW, hipMemcpyDeviceToDevice, st));
  // 2. normals + valid flags
  hipLaunchKernelGGL(k_dp_normals, grid, blk, 0, st, H, W, cfg->normal_model, d->pcd, in->color, d->nrm, d->flag);
  // 3. index_map = exclusive scan of the flags; total = last index + last flag
  size_t bytes = 0;
  DCHK(rocprim::exclusive_scan(nullptr, bytes, d->flag, d->idx, 0, (size_t)HW, rocprim::plus<int32_t>(), st));
  if (bytes > d->cap_tmp) {
    if (d->tmp) DCHK(hipFree(d->tmp));
    d->tmp = nullptr;
    d->cap_tmp = 0;
    DCHK(hipMalloc(&d->tmp, bytes));
    d->cap_tmp = bytes;
  }
  DCHK(rocprim::exclusive_scan(d->tmp, bytes, d->flag, d->idx, 0, (size_t)HW, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(k_dp_gather, grid, blk, 0, st, *cfg, *in, *out, ssim, d->pcd, d->nrm, d->flag, d->idx);
  int32_t last[2];
  DCHK(hipMemcpyAsync(&last[0], d->idx + HW - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  DCHK(hipMemcpyAsync(&last[1], d->flag + HW - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  DCHK(hipStreamSynchronize(st));
  const int T = last[0] + last[1];
  if (n_valid_host) *n_valid_host = T;
  // 4. distance to the class boundaries
  if (in->seg && out->dist2edge && out->points && out->seg && T > 0) {
    slm_gf_semantic sem{};
    sem.num_classes = cfg->num_classes;
    sem.img_seg = in->seg;
    int32_t off[SLM_MAX_CLASSES + 1];
    DCHK(sem_extract_edges(d->sem, sem, H, W, off, st));
    hipLaunchKernelGGL(k_dp_dist2edge, dim3((T + 255) / 256), blk, 0, st, *cfg, T, out->points, out->seg,
                       d->sem.edge_xy, off[0], off[1], off[2], off[3], off[4], out->dist2edge);
  }
  DCHK(hipGetLastError());
  return SLM_OK;
}

}  // extern "C"
