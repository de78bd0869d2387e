// slm_misc.hip -- LM loop state kernels (accept/reject on the device), Surfels.update,
// and the KNN feeder.
#include "slm_common.h"

// ---------------------------------------------------------------------------------
// beta <- identity, state <- initial (reference super/LM.py:81-91)
__global__ void __launch_bounds__(256) k_init_slot(const FrameDev* __restrict__ frames, int slot,
                                                    double u0, double v, double minimal_loss0,
                                                    int num_iterations) {
  const FrameDev& fd = frames[slot];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < fd.f.J) {
    double* b = fd.beta + 7 * t;
    b[0] = 1.0;
#pragma unroll
    for (int c = 1; c < 7; ++c) b[c] = 0.0;
    const double ident[7] = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    pack_node(fd.node_pk + (size_t)SLM_NPK * t, ident, ld_state3(frame_in(fd).ed_points, (size_t)t, fd.f.state_f64));
  }
  if (t < num_iterations) {
    slm_iter_record r;
    r.loss = 0.0;
    r.u = 0.0;
    r.accepted = 0;
    r.status = SLM_ITER_NOT_RUN;
    r.M_grad = 0;
    r.M_loss = 0;
    fd.rec[t] = r;
  }
  if (t == 0) {
    LMState s;
    s.u = u0;
    s.v = v;
    s.minimal_loss = minimal_loss0;
    s.iter = 0;
    s.stopped = 0;
    s.m_grad = 0;
    s.m_loss = 0;
    s.chol_fail = 0;
    s.m_grad_local = 0;
    s.eval_valid = 0;
    s.m_eval = 0;
    s.eval_acc = 0;
    *fd.st = s;
  }
}

// Start of an iteration: zero the band, rhs and counters.  grid = (blocks, n_frames)
__global__ void __launch_bounds__(256) k_iter_begin(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped) return;
  const size_t nband = (size_t)fd.nt * (fd.wb + 1) * SLM_NB * SLM_NB;
  const size_t nrhs = (size_t)fd.nt * SLM_NB;
  double2* b2 = reinterpret_cast<double2*>(fd.band.get());
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nband / 2;
       e += (size_t)gridDim.x * blockDim.x)
    b2[e] = make_double2(0.0, 0.0);
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nrhs;
       e += (size_t)gridDim.x * blockDim.x)
    fd.rhs[e] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    fd.st->m_grad = 0;
    fd.st->chol_fail = 0;
  }
}

// End of an iteration (reference super/LM.py:99-117): reduce the loss partials in a
// fixed order, then accept (u /= v, beta += delta) or reject (u *= v, beta kept).
// A failed factorisation stops the loop with beta unchanged.  grid = (1, n_frames), 1024 thr.
// Records are kept for the first n_rec iterations after the bind (the capacity of fd.rec): running again
// without binding continues the loop from the current state and records nothing more.
// eval_pass: the loss pass of this iteration was k_data_eval (it wrote the slot's evaluation buffer at the trial point);
// 0 when the batch took k_data_loss (one slot without a tuple-sorted plan sends all of them there): ev_rc then still
// holds an OLDER evaluation and must not be marked valid.
__global__ void __launch_bounds__(1024) k_accept(const FrameDev* __restrict__ frames, int phase_test,
                                                  int n_reg_part, int n_rec, int* __restrict__ reuse, int eval_pass) {
  __shared__ double sm[16];
  __shared__ int s_accept;
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound) return;
  LMState* st = fd.st;
  if (st->stopped) return;
  const int it = st->iter;
  if (st->chol_fail) {
    if (threadIdx.x == 0) {
      if (it < n_rec) {
        slm_iter_record r = fd.rec[it];
        r.status = st->chol_fail == 2 ? SLM_ITER_SOLVER_TIMEOUT : SLM_ITER_SOLVER_FAILED;
        r.u = st->u;
        r.M_grad = st->m_grad;
        fd.rec[it] = r;
      }
      st->stopped = 1;
    }
    return;
  }
  double ls = 0.0, cnt = 0.0;
  for (int b = threadIdx.x; b < fd.n_loss_part; b += blockDim.x) {
    ls += fd.loss_part[2 * b];
    cnt += fd.loss_part[2 * b + 1];
  }
  const double* reg = fd.loss_part + 2 * (size_t)fd.n_loss_part;
  for (int b = threadIdx.x; b < n_reg_part; b += blockDim.x) ls += reg[2 * b] + reg[2 * b + 1];
  const double loss = block_sum(ls, sm);
  const double m = block_sum(cnt, sm);
  if (threadIdx.x == 0) {
    int acc = 1;
    const double u_used = st->u;
    if (phase_test) {
      if (loss < st->minimal_loss) {
        st->minimal_loss = loss;
        st->u = u_used / st->v;
      } else {
        acc = 0;
        st->u = u_used * st->v;
      }
    }
    slm_iter_record r;
    r.loss = loss;
    r.u = u_used;
    r.accepted = acc;
    r.status = SLM_ITER_OK;
    r.M_grad = st->m_grad;
    r.M_loss = (int)m;
    if (it < n_rec) fd.rec[it] = r;
    st->iter = it + 1;
    s_accept = acc;
    if (reuse) reuse[blockIdx.y] = acc ? 0 : 1;   // rejected: the next Jacobian pass would repeat this one (k_data_gram)
    st->eval_valid = eval_pass ? acc : 0;         // the loss pass evaluated the TRIAL point: the current beta only if accepted
  }
  __syncthreads();
  if (s_accept) {
    for (int e = threadIdx.x; e < fd.P; e += blockDim.x) {
      const double b = fd.beta[e] + fd.delta[e];
      fd.beta[e] = b;
      fd.node_pk[(size_t)SLM_NPK * (e / 7) + e % 7] = b;
    }
  }
}

// node_pk <- beta (after slm_set_beta); grid = (ceil(J/256), 1)
__global__ void __launch_bounds__(256) k_pack_nodes(const FrameDev* __restrict__ frames, int slot) {
  const FrameDev& fd = frames[slot];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= fd.f.J) return;
  double bb[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) bb[c] = fd.beta[7 * j + c];
  pack_node(fd.node_pk + (size_t)SLM_NPK * j, bb, ld_state3(frame_in(fd).ed_points, (size_t)j, fd.f.state_f64));
}

// node_pk_try <- beta + delta: the trial point of the loss pass; grid = (ceil(maxJ/256), n_frames)
__global__ void __launch_bounds__(256) k_make_trial(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped) return;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= fd.f.J) return;
  double bb[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) bb[c] = fd.beta[7 * j + c] + fd.delta[7 * j + c];
  pack_node(fd.node_pk_try + (size_t)SLM_NPK * j, bb, ld_state3(frame_in(fd).ed_points, (size_t)j, fd.f.state_f64));
}

// target points + normals -> interleaved float4 pairs; grid = (ceil(T/256))
__global__ void __launch_bounds__(256) k_pack_target(int T, const float* __restrict__ pts,
                                                      const float* __restrict__ nrm,
                                                      float4* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  out[2 * (size_t)t] = make_float4(pts[3 * t], pts[3 * t + 1], pts[3 * t + 2], 0.f);
  out[2 * (size_t)t + 1] = make_float4(nrm[3 * t], nrm[3 * t + 1], nrm[3 * t + 2], 0.f);
}

// per-pixel form of the same (FrameDev::tgt_px); grid = (ceil(H*W/256))
__global__ void __launch_bounds__(256) k_pack_target_px(int HW, const int* __restrict__ index_map, const uint8_t* __restrict__ valid,
                                                         const float* __restrict__ pts, const float* __restrict__ nrm,
                                                         float4* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int t = index_map[p];
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, valid[p] ? 1.f : 0.f);
  if (t >= 0) {
    a = make_float4(pts[3 * (size_t)t], pts[3 * (size_t)t + 1], pts[3 * (size_t)t + 2], 1.f);
    b = make_float4(nrm[3 * (size_t)t], nrm[3 * (size_t)t + 1], nrm[3 * (size_t)t + 2], b.w);
  }
  out[2 * (size_t)p] = a;
  out[2 * (size_t)p + 1] = b;
}

// Reduce loss partials into out[0..3] = data, arap, rot, matched count (slm_loss).
__global__ void __launch_bounds__(256) k_loss_out(const FrameDev* __restrict__ frames, int slot,
                                                   int n_reg_part, double* __restrict__ out) {
  __shared__ double sm[16];
  const FrameDev& fd = frames[slot];
  double d = 0.0, c = 0.0, a = 0.0, r = 0.0;
  for (int b = threadIdx.x; b < fd.n_loss_part; b += blockDim.x) {
    d += fd.loss_part[2 * b];
    c += fd.loss_part[2 * b + 1];
  }
  const double* reg = fd.loss_part + 2 * (size_t)fd.n_loss_part;
  for (int b = threadIdx.x; b < n_reg_part; b += blockDim.x) {
    a += reg[2 * b];
    r += reg[2 * b + 1];
  }
  d = block_sum(d, sm);
  c = block_sum(c, sm);
  a = block_sum(a, sm);
  r = block_sum(r, sm);
  if (threadIdx.x == 0) {
    out[0] = d;
    out[1] = a;
    out[2] = r;
    out[3] = c;
  }
}

__global__ void k_zero_reg_part(const FrameDev* __restrict__ frames, int n_reg_part) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound) return;
  double* reg = fd.loss_part + 2 * (size_t)fd.n_loss_part;
  for (int b = threadIdx.x; b < 2 * n_reg_part; b += blockDim.x) reg[b] = 0.0;
}

// ---------------------------------------------------------------------------------
// Surfels.update, LM variant (reference super/nodes.py:193-223): in place, on float32 or float64
// arrays (RT).  Note the reference rotates the normals with the 7-wide beta, so the translation b_k
// is ADDED to the rotated normal before blending and normalising (nodes.py:207-213).
template <typename RT>
__global__ void __launch_bounds__(256) k_update_surfels(int N, int K, RT* __restrict__ pts,
                                                         RT* __restrict__ nrm,
                                                         const int* __restrict__ knn_idx,
                                                         const RT* __restrict__ knn_w,
                                                         const RT* __restrict__ ed_pts,
                                                         const double* __restrict__ beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t i3 = 3 * (size_t)i;
  const d3 p = {(double)pts[i3], (double)pts[i3 + 1], (double)pts[i3 + 2]};
  const d3 n0 = {(double)nrm[i3], (double)nrm[i3 + 1], (double)nrm[i3 + 2]};
  d3 T = {0, 0, 0}, Nn = {0, 0, 0};
  // (K = opt.num_neighbors, 1..8; the sums run k = 0, 1, ... like the reference's sum over dim 1)
  for (int k = 0; k < K; ++k) {
    const int id = knn_idx[(size_t)K * i + k];
    const double wk = (double)knn_w[(size_t)K * i + k];
    const double* b = beta + 7 * id;
    const d3 g = {(double)ed_pts[3 * id], (double)ed_pts[3 * id + 1], (double)ed_pts[3 * id + 2]};
    const d3 qv = {b[1], b[2], b[3]};
    d3 t = quat_apply(b[0], qv, p - g);
    t = {t.x + b[4] + g.x, t.y + b[5] + g.y, t.z + b[6] + g.z};
    T = {T.x + wk * t.x, T.y + wk * t.y, T.z + wk * t.z};
    d3 rn = quat_apply(b[0], qv, n0);
    rn = {rn.x + b[4], rn.y + b[5], rn.z + b[6]};
    Nn = {Nn.x + wk * rn.x, Nn.y + wk * rn.y, Nn.z + wk * rn.z};
  }
  const double nl = fmax(sqrt(dot(Nn, Nn)), 1e-12);   // F.normalize eps
  pts[i3] = (RT)T.x;
  pts[i3 + 1] = (RT)T.y;
  pts[i3 + 2] = (RT)T.z;
  nrm[i3] = (RT)(Nn.x / nl);
  nrm[i3 + 1] = (RT)(Nn.y / nl);
  nrm[i3 + 2] = (RT)(Nn.z / nl);
}

// must run AFTER k_update_surfels (which reads the old node positions)
template <typename RT>
__global__ void __launch_bounds__(256) k_update_nodes(int J, RT* __restrict__ ed_pts,
                                                       RT* __restrict__ ed_nrm,
                                                       const double* __restrict__ beta) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  const double* b = beta + 7 * j;
  const d3 n0 = {(double)ed_nrm[3 * j], (double)ed_nrm[3 * j + 1], (double)ed_nrm[3 * j + 2]};
  const d3 rn = quat_apply(b[0], {b[1], b[2], b[3]}, n0);
  const double nl = fmax(sqrt(dot(rn, rn)), 1e-12);
  ed_pts[3 * j] = (RT)((double)ed_pts[3 * j] + b[4]);
  ed_pts[3 * j + 1] = (RT)((double)ed_pts[3 * j + 1] + b[5]);
  ed_pts[3 * j + 2] = (RT)((double)ed_pts[3 * j + 2] + b[6]);
  ed_nrm[3 * j] = (RT)(rn.x / nl);
  ed_nrm[3 * j + 1] = (RT)(rn.y / nl);
  ed_nrm[3 * j + 2] = (RT)(rn.z / nl);
}

// ---------------------------------------------------------------------------------
// Brute-force KNN (replaces pytorch3d.ops.knn_points as called by utils/utils.py:217):
// node tiles staged through LDS, every thread keeps its K best in registers
// (squared L2 in f64, ascending, ties -> lowest index because nodes are visited in order
// and a candidate replaces only on strictly smaller distance).
#define KNN_MAXK 9
#define KNN_TILE 1024
__global__ void __launch_bounds__(256) k_knn(int Nq, int Nn, int K, int skip_self,
                                              const float* __restrict__ q, const float* __restrict__ nodes,
                                              int* __restrict__ idx_out, float* __restrict__ dist_out) {
  __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int KK = K + (skip_self ? 1 : 0);
  double bd[KNN_MAXK];
  int bi[KNN_MAXK];
#pragma unroll
  for (int k = 0; k < KNN_MAXK; ++k) {
    bd[k] = 1e300;
    bi[k] = -1;
  }
  double px = 0, py = 0, pz = 0;
  if (i < Nq) {
    px = q[3 * i];
    py = q[3 * i + 1];
    pz = q[3 * i + 2];
  }
  for (int base = 0; base < Nn; base += KNN_TILE) {
    const int cnt = min(KNN_TILE, Nn - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
      sx[t] = nodes[3 * (base + t)];
      sy[t] = nodes[3 * (base + t) + 1];
      sz[t] = nodes[3 * (base + t) + 2];
    }
    __syncthreads();
    if (i < Nq) {
      for (int t = 0; t < cnt; ++t) {
        const double dx = px - (double)sx[t], dy = py - (double)sy[t], dz = pz - (double)sz[t];
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < bd[KNN_MAXK - 1]) {
          // insertion into the sorted list (static indexing keeps it in registers)
          double cd = d2;
          int ci = base + t;
#pragma unroll
          for (int k = 0; k < KNN_MAXK; ++k) {
            if (cd < bd[k]) {
              const double td = bd[k];
              const int ti = bi[k];
              bd[k] = cd;
              bi[k] = ci;
              cd = td;
              ci = ti;
            }
          }
        }
      }
    }
  }
  if (i < Nq) {
    const int off = skip_self ? 1 : 0;
    for (int k = 0; k < K; ++k) {
      idx_out[i * K + k] = bi[k + off];
      dist_out[i * K + k] = (float)sqrt(bd[k + off]);
    }
  }
  (void)KK;
}

// softmax(exp(-dist/radius)) weights + stability test (reference super/nodes.py:166,182,191)
__global__ void __launch_bounds__(256) k_knn_weights(int Nq, int K, int radius_mode,
                                                      const int* __restrict__ idx,
                                                      const float* __restrict__ dist,
                                                      const float* __restrict__ radii,
                                                      float* __restrict__ w_out,
                                                      uint8_t* __restrict__ stable_io) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nq) return;
  double e[KNN_MAXK];
  double emax = -1e300;
  bool any_in = false;
  for (int k = 0; k < K; ++k) {
    const double r = (double)radii[radius_mode == 0 ? idx[i * K + k] : i];
    const double d = (double)dist[i * K + k];
    any_in = any_in || (d <= r);
    e[k] = exp(-d / r);
    emax = fmax(emax, e[k]);
  }
  double s = 0.0;
  for (int k = 0; k < K; ++k) {
    e[k] = exp(e[k] - emax);
    s += e[k];
  }
  for (int k = 0; k < K; ++k) w_out[i * K + k] = (float)(e[k] / s);
  if (stable_io && !any_in) stable_io[i] = 0;
}

// float64 variant of the feeder with the Semantic-SuPer branches (find_knn with num_classes,
// utils/utils.py:223-242: a query only sees the nodes of its own class; weights with the
// Jensen-Shannon factor, super/nodes.py:183-189).  counter[0] counts queries that found fewer than
// K (+ self) nodes of their class -- the reference asserts on those.
__global__ void __launch_bounds__(256) k_knn64(int Nq, int Nn, int K, int skip_self, const double* __restrict__ q,
                                                const double* __restrict__ nodes, const int* __restrict__ q_seg,
                                                const int* __restrict__ node_seg, int* __restrict__ idx_out,
                                                double* __restrict__ dist_out, int* __restrict__ counter) {
  __shared__ double sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  __shared__ int sc[KNN_TILE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool by_class = q_seg != nullptr && node_seg != nullptr;
  double bd[KNN_MAXK];
  int bi[KNN_MAXK];
#pragma unroll
  for (int k = 0; k < KNN_MAXK; ++k) {
    bd[k] = 1e300;
    bi[k] = -1;
  }
  double px = 0, py = 0, pz = 0;
  int cls = 0;
  if (i < Nq) {
    px = q[3 * (size_t)i];
    py = q[3 * (size_t)i + 1];
    pz = q[3 * (size_t)i + 2];
    if (by_class) cls = q_seg[i];
  }
  for (int base = 0; base < Nn; base += KNN_TILE) {
    const int cnt = min(KNN_TILE, Nn - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
      sx[t] = nodes[3 * (size_t)(base + t)];
      sy[t] = nodes[3 * (size_t)(base + t) + 1];
      sz[t] = nodes[3 * (size_t)(base + t) + 2];
      if (by_class) sc[t] = node_seg[base + t];
    }
    __syncthreads();
    if (i < Nq) {
      for (int t = 0; t < cnt; ++t) {
        if (by_class && sc[t] != cls) continue;
        const double dx = px - sx[t], dy = py - sy[t], dz = pz - sz[t];
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < bd[KNN_MAXK - 1]) {
          double cd = d2;
          int ci = base + t;
#pragma unroll
          for (int k = 0; k < KNN_MAXK; ++k) {
            if (cd < bd[k]) {
              const double td = bd[k];
              const int ti = bi[k];
              bd[k] = cd;
              bi[k] = ci;
              cd = td;
              ci = ti;
            }
          }
        }
      }
    }
  }
  if (i < Nq) {
    const int off = skip_self ? 1 : 0;
    bool short_of = false;
    for (int k = 0; k < K; ++k) {
      idx_out[(size_t)i * K + k] = bi[k + off];
      dist_out[(size_t)i * K + k] = sqrt(bd[k + off]);
      short_of = short_of || bi[k + off] < 0;
    }
    if (short_of && counter) atomicAdd(counter, 1);
  }
}

// weights in float64; q_seg_conf / node_seg_conf (C columns) switch on the Jensen-Shannon factor
__global__ void __launch_bounds__(256) k_knn_weights64(int Nq, int K, int radius_mode, const int* __restrict__ idx,
                                                        const double* __restrict__ dist, const double* __restrict__ radii,
                                                        int C, const double* __restrict__ q_seg_conf,
                                                        const double* __restrict__ node_seg_conf,
                                                        double* __restrict__ w_out, uint8_t* __restrict__ stable_io) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nq) return;
  double e[KNN_MAXK];
  double emax = -1e300;
  bool any_in = false;
  for (int k = 0; k < K; ++k) {
    const int j = idx[(size_t)i * K + k];
    const double r = radii[radius_mode == 0 ? j : i];
    const double d = dist[(size_t)i * K + k];
    any_in = any_in || (d <= r);
    e[k] = exp(-d / r);
    if (C > 0) {
      const double* P = node_seg_conf + (size_t)C * j;
      const double* Q = q_seg_conf + (size_t)C * i;
      double a = 0.0, b = 0.0;
      for (int c = 0; c < C; ++c) {
        const double M = 0.5 * (P[c] + Q[c]);
        a += P[c] * log(P[c] / (M + 1e-13) + 1e-13);
        b += Q[c] * log(Q[c] / (M + 1e-13) + 1e-13);
      }
      e[k] = sqrt(exp(-0.5 * (a + b))) * sqrt(e[k]);
    }
    emax = fmax(emax, e[k]);
  }
  double s = 0.0;
  for (int k = 0; k < K; ++k) {
    e[k] = exp(e[k] - emax);
    s += e[k];
  }
  for (int k = 0; k < K; ++k) w_out[(size_t)i * K + k] = e[k] / s;
  if (stable_io && !any_in) stable_io[i] = 0;
}

// ---- host launchers --------------------------------------------------------------
void launch_init_slot(const FrameDev* frames_dev, int slot, int J, const slm_config& cfg,
                      hipStream_t st) {
  int n = J > cfg.num_iterations ? J : cfg.num_iterations;
  if (n < 1) n = 1;
  hipLaunchKernelGGL(k_init_slot, dim3((n + 255) / 256), dim3(256), 0, st, frames_dev, slot, cfg.u0,
                     cfg.v, cfg.minimal_loss0, cfg.num_iterations);
}

void launch_pack_nodes(const FrameDev* frames_dev, int slot, int J, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_nodes, dim3((J + 255) / 256), dim3(256), 0, st, frames_dev, slot);
}

void launch_make_trial(const FrameDev* frames_dev, int n_frames, int maxJ, hipStream_t st) {
  hipLaunchKernelGGL(k_make_trial, dim3((maxJ + 255) / 256, n_frames), dim3(256), 0, st, frames_dev);
}

void launch_pack_target(int T, const float* pts, const float* nrm, float4* out, hipStream_t st) {
  if (T > 0) hipLaunchKernelGGL(k_pack_target, dim3((T + 255) / 256), dim3(256), 0, st, T, pts, nrm, out);
}

void launch_pack_target_px(int HW, const int* index_map, const uint8_t* valid, const float* pts, const float* nrm, float4* out,
                           hipStream_t st) {
  if (HW > 0) hipLaunchKernelGGL(k_pack_target_px, dim3((HW + 255) / 256), dim3(256), 0, st, HW, index_map, valid, pts, nrm, out);
}

void launch_iter_begin(const FrameDev* frames_dev, int n_frames, hipStream_t st) {
  hipLaunchKernelGGL(k_iter_begin, dim3(1024, n_frames), dim3(256), 0, st, frames_dev);
}

void launch_accept(const FrameDev* frames_dev, int n_frames, int phase_test, int n_reg_part, int n_rec,
                   hipStream_t st, int* reuse, int eval_pass) {
  hipLaunchKernelGGL(k_accept, dim3(1, n_frames), dim3(1024), 0, st, frames_dev, phase_test,
                     n_reg_part, n_rec, reuse, eval_pass);
}

void launch_loss_out(const FrameDev* frames_dev, int slot, int n_reg_part, double* out,
                     hipStream_t st) {
  hipLaunchKernelGGL(k_loss_out, dim3(1), dim3(256), 0, st, frames_dev, slot, n_reg_part, out);
}

void launch_zero_reg_part(const FrameDev* frames_dev, int n_frames, int n_reg_part, hipStream_t st) {
  hipLaunchKernelGGL(k_zero_reg_part, dim3(1, n_frames), dim3(256), 0, st, frames_dev, n_reg_part);
}

template <typename RT>
static void launch_update_t(int N, int J, int K, RT* pts, RT* nrm, const int* knn_idx, const RT* knn_w,
                            RT* ed_pts, RT* ed_nrm, const double* beta, hipStream_t st) {
  if (N > 0)
    hipLaunchKernelGGL(k_update_surfels<RT>, dim3((N + 255) / 256), dim3(256), 0, st, N, K, pts, nrm,
                       knn_idx, knn_w, (const RT*)ed_pts, beta);
  if (J > 0)
    hipLaunchKernelGGL(k_update_nodes<RT>, dim3((J + 255) / 256), dim3(256), 0, st, J, ed_pts, ed_nrm,
                       beta);
}
void launch_update(int N, int J, int K, float* pts, float* nrm, const int* knn_idx, const float* knn_w,
                   float* ed_pts, float* ed_nrm, const double* beta, hipStream_t st) {
  launch_update_t<float>(N, J, K, pts, nrm, knn_idx, knn_w, ed_pts, ed_nrm, beta, st);
}
void launch_update64(int N, int J, int K, double* pts, double* nrm, const int* knn_idx, const double* knn_w,
                     double* ed_pts, double* ed_nrm, const double* beta, hipStream_t st) {
  launch_update_t<double>(N, J, K, pts, nrm, knn_idx, knn_w, ed_pts, ed_nrm, beta, st);
}

void launch_knn(int Nq, int Nn, int K, int skip_self, const float* q, const float* nodes, int* idx,
                float* dist, hipStream_t st) {
  if (Nq <= 0) return;
  hipLaunchKernelGGL(k_knn, dim3((Nq + 255) / 256), dim3(256), 0, st, Nq, Nn, K, skip_self, q, nodes,
                     idx, dist);
}

void launch_knn64(int Nq, int Nn, int K, int skip_self, const double* q, const double* nodes, const int* q_seg,
                  const int* node_seg, int* idx, double* dist, int* counter, hipStream_t st) {
  if (Nq <= 0) return;
  hipLaunchKernelGGL(k_knn64, dim3((Nq + 255) / 256), dim3(256), 0, st, Nq, Nn, K, skip_self, q, nodes, q_seg,
                     node_seg, idx, dist, counter);
}

void launch_knn_weights64(int Nq, int K, int radius_mode, const int* idx, const double* dist, const double* radii,
                          int C, const double* q_conf, const double* node_conf, double* w, uint8_t* stable,
                          hipStream_t st) {
  if (Nq <= 0) return;
  hipLaunchKernelGGL(k_knn_weights64, dim3((Nq + 255) / 256), dim3(256), 0, st, Nq, K, radius_mode, idx, dist,
                     radii, C, q_conf, node_conf, w, stable);
}

void launch_knn_weights(int Nq, int K, int radius_mode, const int* idx, const float* dist,
                        const float* radii, float* w, uint8_t* stable, hipStream_t st) {
  if (Nq <= 0) return;
  hipLaunchKernelGGL(k_knn_weights, dim3((Nq + 255) / 256), dim3(256), 0, st, Nq, K, radius_mode,
                     idx, dist, radii, w, stable);
}
