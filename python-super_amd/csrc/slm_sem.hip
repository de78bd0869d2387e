// slm_sem.hip -- Semantic-SuPer terms of the reference's GraphFit (autograd) path:
//
//   k_edge_flags / sem_extract_edges   class-boundary pixels of inputs[("seg",0)], once per frame
//                                      (find_edge_region with kernel 3, utils/utils.py:276-301, as
//                                      called at super/deform_mesh.py:149-165), row-major order
//   k_gf_morph                         semantic-boundary morphing term (super/deform_mesh.py:126-194):
//                                      surfels that project onto a pixel of another class are pulled
//                                      towards the 2 nearest boundary pixels of their own class
//
// The soft / hard segmentation weight of the point-to-plane term lives in k_gf_data (slm_gf.hip).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "slm_sem.h"

namespace {

// one thread per pixel: is it a boundary pixel of its own class?
__global__ void __launch_bounds__(256) k_edge_flags(int H, int W, int C, const int32_t* __restrict__ seg,
                                                     uint8_t* __restrict__ flags, int32_t* __restrict__ counts) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= H * W) return;
  const int y = pix / W, x = pix % W;
  const int c0 = seg[pix];
  bool edge = false;
  // ignore_img_edge: kernel (3) rows / columns at every image side; the margin-1 test is implied
  if (c0 >= 0 && c0 < C && y >= 3 && y < H - 3 && x >= 3 && x < W - 3) {
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) edge = edge || (seg[(y + dy) * W + (x + dx)] != c0);
  }
  for (int c = 0; c < C; ++c) flags[(size_t)c * H * W + pix] = (edge && c == c0) ? 1 : 0;
  if (edge) atomicAdd(&counts[c0], 1);
}

__global__ void __launch_bounds__(256) k_edge_pack(int n, int HW, int W, const int32_t* __restrict__ sel,
                                                    float2* __restrict__ xy) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int pix = sel[e] % HW;
  xy[e] = make_float2((float)(pix % W), (float)(pix / W));
}

template <typename T>
hipError_t grow(T*& p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  if (p) {
    hipError_t e = hipFree(p);
    if (e != hipSuccess) return e;
    p = nullptr;
    cap = 0;
  }
  hipError_t e = hipMalloc((void**)&p, need * sizeof(T));
  if (e == hipSuccess) cap = need;
  return e;
}

}  // namespace

#define SCHK(expr)                   \
  do {                               \
    hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return e_; \
  } while (0)

hipError_t sem_extract_edges(SemScratch& sc, const slm_gf_semantic& sem, int H, int W, int32_t* edge_off,
                             hipStream_t st) {
  const int C = sem.num_classes;
  const size_t HW = (size_t)H * W, n = HW * C;
  SCHK(grow(sc.flags, sc.cap_flags, n));
  if (!sc.counts) SCHK(hipMalloc((void**)&sc.counts, sizeof(int32_t) * (SLM_MAX_CLASSES + 1)));
  SCHK(hipMemsetAsync(sc.counts, 0, sizeof(int32_t) * (SLM_MAX_CLASSES + 1), st));
  hipLaunchKernelGGL(k_edge_flags, dim3((HW + 255) / 256), dim3(256), 0, st, H, W, C, sem.img_seg, sc.flags,
                     sc.counts);
  int32_t h[SLM_MAX_CLASSES + 1];
  SCHK(hipMemcpyAsync(h, sc.counts, sizeof(h), hipMemcpyDeviceToHost, st));
  SCHK(hipStreamSynchronize(st));
  edge_off[0] = 0;
  for (int c = 0; c < SLM_MAX_CLASSES; ++c) edge_off[c + 1] = edge_off[c] + (c < C ? h[c] : 0);
  const int total = edge_off[C];
  SCHK(grow(sc.sel, sc.cap_sel, (size_t)total + 1));
  SCHK(grow(sc.edge_xy, sc.cap_edge, (size_t)total + 1));
  if (total == 0) return hipSuccess;
  size_t bytes = 0;
  rocprim::counting_iterator<int32_t> first(0);
  SCHK(rocprim::select(nullptr, bytes, first, sc.flags, sc.sel, sc.counts + SLM_MAX_CLASSES, n, st));
  if (bytes > sc.cap_tmp) {
    if (sc.tmp) SCHK(hipFree(sc.tmp));
    sc.tmp = nullptr;
    sc.cap_tmp = 0;
    SCHK(hipMalloc(&sc.tmp, bytes));
    sc.cap_tmp = bytes;
  }
  SCHK(rocprim::select(sc.tmp, bytes, first, sc.flags, sc.sel, sc.counts + SLM_MAX_CLASSES, n, st));
  hipLaunchKernelGGL(k_edge_pack, dim3((total + 255) / 256), dim3(256), 0, st, total, (int)HW, W, sc.sel,
                     sc.edge_xy);
  return hipGetLastError();
}

hipError_t sem_ensure_morph(SemScratch& sc, int N) { return grow(sc.morph_g, sc.cap_morph, (size_t)N + 1); }

void sem_free(SemScratch& sc) {
  void* ptrs[] = {sc.flags, sc.sel, sc.counts, sc.tmp, sc.edge_xy, sc.morph_g};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  sc = SemScratch();
}

// grid = (ceil(maxN/256), n_frames)
__global__ void __launch_bounds__(256) k_gf_morph(GfSlot* __restrict__ slots) {
  __shared__ double sm[16];
  GfSlotDev& s = gf_dev(slots)[blockIdx.y];
  if (!s.bound || !s.sem_bound) return;
  const FrameIn& f = s.f.base;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double li_sum = 0.0, kept = 0.0;
  // candidates of this workgroup (surfels whose projected class differs from their own), searched cooperatively below
  __shared__ int n_cand;
  __shared__ double c_x[256], c_y[256], c_d1[256], c_d2[256];
  __shared__ int c_e0[256], c_e1[256], c_i1[256], c_i2[256];
  if (threadIdx.x == 0) n_cand = 0;
  __syncthreads();
  int my_cand = -1, my_W = 0, my_H = 0;
  double my_x = 0.0, my_y = 0.0;
  if (i < f.N) {
    if (i >= s.shard_lo && i < s.shard_hi && (!s.f.sf_stable || s.f.sf_stable[i])) {
      struct { d3 P; } k;
      if (f.K == SLM_K) {   // (the default: the instantiation the other kernels use, operation for operation)
        GfSkin k4;
        gf_skin(s, i, k4);
        k.P = k4.P;
      } else {
        k.P = gf_skin_pos(s, i);
      }
      const int H = f.H, W = f.W, C = s.sem.num_classes;
      const double Ze = k.P.z + 1e-8;
      const double x = k.P.x * (double)f.fx / Ze + (double)f.cx, y = k.P.y * (double)f.fy / Ze + (double)f.cy;
      // F.grid_sample(seg_conf, grid) with bilinear taps, zero padding, align_corners=False
      const double gx = x / (double)W * 2.0 - 1.0, gy = y / (double)H * 2.0 - 1.0;
      const double ix = ((gx + 1.0) * (double)W - 1.0) / 2.0, iy = ((gy + 1.0) * (double)H - 1.0) / 2.0;
      const double x0 = floor(ix), y0 = floor(iy);
      const double wnw = (x0 + 1.0 - ix) * (y0 + 1.0 - iy), wne = (ix - x0) * (y0 + 1.0 - iy);
      const double wsw = (x0 + 1.0 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
      const bool inx0 = x0 >= 0.0 && x0 <= (double)(W - 1), inx1 = x0 + 1.0 >= 0.0 && x0 + 1.0 <= (double)(W - 1);
      const bool iny0 = y0 >= 0.0 && y0 <= (double)(H - 1), iny1 = y0 + 1.0 >= 0.0 && y0 + 1.0 <= (double)(H - 1);
      int best = 0;
      double bestv = 0.0;
      if (gx > -4.0 && gx < 4.0 && gy > -4.0 && gy < 4.0) {   // far outside: all taps are padding
        const int xi = (int)x0, yi = (int)y0;
        for (int c = 0; c < C; ++c) {
          const float* img = s.sem.img_seg_conf + (size_t)c * H * W;
          double v = 0.0;
          if (inx0 && iny0) v += (double)img[yi * W + xi] * wnw;
          if (inx1 && iny0) v += (double)img[yi * W + xi + 1] * wne;
          if (inx0 && iny1) v += (double)img[(yi + 1) * W + xi] * wsw;
          if (inx1 && iny1) v += (double)img[(yi + 1) * W + xi + 1] * wse;
          if (c == 0 || v > bestv) {
            bestv = v;
            best = c;
          }
        }
      }
      const int cls = s.sem.sf_seg[i];
      const bool val = best != cls && gx > -1.0 && gx < 1.0 && gy > -1.0 && gy < 1.0;
      if (val && cls >= 0 && cls < C) {
        const int e0 = s.edge_off[cls], e1 = s.edge_off[cls + 1];
        if (e1 - e0 >= 2) {      // a single boundary pixel has no 2nd neighbour: treated as no boundary
          s.terms[7] = 1.0;      // the class contributes a list entry (same value from every writer)
          // a candidate: its search is done by the WHOLE workgroup below
          const int k = atomicAdd(&n_cand, 1);
          c_x[k] = x;
          c_y[k] = y;
          c_e0[k] = e0;
          c_e1[k] = e1;
          my_cand = k;
          my_x = x;
          my_y = y;
          my_W = W;
          my_H = H;
        }
      }
    }
  }
  __syncthreads();
  // ---- the two nearest boundary pixels of every candidate's class: the mismatching surfels are a thin band along the class
  // boundaries (a few per cent), so one lane per surfel scanning ~1 000-2 000 boundary pixels left 60 lanes of its wave idle for
  // the whole scan (534 us per launch at C4, 80 % of the Semantic-SuPer step).  The workgroup's WAVES take its candidates one
  // after the other (wave w the candidates w, w + 4, ...): 64 lanes scan the class's list strided, each keeps its two nearest
  // in (distance, list position) order -- the order of the sequential scan with strict <, i.e. find_knn's lowest index on
  // ties -- and a butterfly merge gives the candidate's answer to lane 0.  (First form of round 6: the whole workgroup per
  // candidate, 256 lanes + a four-way merge through LDS behind two barriers per candidate -- the butterfly and the barriers
  // were issued by four waves for one candidate: 68 us per launch at C4.)
  {
    const int nc = n_cand;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = w; k < nc; k += 4) {
      const double x = c_x[k], y = c_y[k];
      const int e0 = c_e0[k], e1 = c_e1[k];
      double d1 = 1e300, d2 = 1e300;
      int i1 = 0x7fffffff, i2 = 0x7fffffff;
      for (int e = e0 + l; e < e1; e += 64) {
        const float2 q = s.edge_xy[e];
        const double dx = x - (double)q.x, dy = y - (double)q.y;
        const double d = dx * dx + dy * dy;
        if (d < d1) {
          d2 = d1; i2 = i1;
          d1 = d; i1 = e;
        } else if (d < d2) {
          d2 = d; i2 = e;
        }
      }
      // merge of two sorted pairs under (distance, index) order
      auto before = [](double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); };
      auto merge = [&](double od1, int oi1, double od2, int oi2) {
        // the two smallest of {(d1,i1) <= (d2,i2)} and {(od1,oi1) <= (od2,oi2)}
        if (before(od1, oi1, d1, i1)) {
          // other's first leads: second is min(mine first, other's second)
          if (before(od2, oi2, d1, i1)) { d2 = od2; i2 = oi2; }
          else { d2 = d1; i2 = i1; }
          d1 = od1; i1 = oi1;
        } else if (before(od1, oi1, d2, i2)) {
          d2 = od1; i2 = oi1;
        }
      };
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double od1 = __shfl_xor(d1, off, 64), od2 = __shfl_xor(d2, off, 64);
        const int oi1 = __shfl_xor(i1, off, 64), oi2 = __shfl_xor(i2, off, 64);
        merge(od1, oi1, od2, oi2);
      }
      if (l == 0) {
        c_d1[k] = d1; c_d2[k] = d2;
        c_i1[k] = i1; c_i2[k] = i2;
      }
    }
  }
  __syncthreads();
  if (i < f.N) {
    double2 g = make_double2(0.0, 0.0);
    if (my_cand >= 0) {
      const double x = my_x, y = my_y, d1 = c_d1[my_cand], d2 = c_d2[my_cand];
      const float2 p1 = s.edge_xy[c_i1[my_cand]], p2 = s.edge_xy[c_i2[my_cand]];
      // drop surfels closer to the image border than to the class boundary
      const double dte = fmin(fmin(fmin(x, y), (double)my_W - x), (double)my_H - y);
      const bool ok = !(sqrt(d1) > dte || sqrt(d2) > dte);
      const double li = (d1 + d2) / 2.0;
      if (ok && li > 15.0) {
        li_sum = li;
        kept = 1.0;
        g = make_double2(-(((double)p1.x - x) + ((double)p2.x - x)), -(((double)p1.y - y) + ((double)p2.y - y)));
      }
    }
    s.morph_g[i] = g;
  }
  const double a = block_sum(li_sum, sm), b = block_sum(kept, sm);
  if (threadIdx.x == 0 && b != 0.0) {   // (spread block partials, slm_gf.h: entries 14 / 15; k_gf_fold sums them into terms[5] / [6])
    double* part = s.terms.get() + SLM_GF_NTERMS + 16 * (blockIdx.x % GF_NCOPY);
    atomic_add_f64(part + 14, a);
    atomic_add_f64(part + 15, b);
  }
}

void launch_gf_morph(GfSlot* slots, int n_frames, int maxN, hipStream_t st) {
  if (maxN <= 0) return;
  hipLaunchKernelGGL(k_gf_morph, dim3((maxN + 255) / 256, n_frames), dim3(256), 0, st, slots);
}
