// SE(3)-equivariant structure module kernels (gfx950), fp32 throughout -- the reference forces
// fp32 here (se3_modules.py:164).  The reference builds a DGL graph with a data-dependent edge
// count (rf.py:823-862, host sync at torch.where); here the graph lives in fixed-capacity device
// buffers: a dense [B,L,L] mask, an edge list compacted on the device in the reference's row-major
// order, and a dense edge-id map that gives each destination node its incoming edges in a fixed
// order (deterministic softmax / scatter-sum, no atomics, no host round trip).
#include "common.h"

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// kNN mask: block per (b,i).  rank(j) = #{j' : d[j'] < d[j] or (d[j'] == d[j] and j' < j)}
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_mask_kernel(const float* xyz, const int64_t* aa_idx, uint8_t* mask, int L,
                                                       int k, int kmin) {
  extern __shared__ float dist[];  // [L]
  const int bi = blockIdx.x, b = bi / L, i = bi % L;
  const float* ci = xyz + ((int64_t)bi * 3 + 1) * 3;
  const float cx = ci[0], cy = ci[1], cz = ci[2];
  for (int j = threadIdx.x; j < L; j += 256) {
    const float* cj = xyz + (((int64_t)b * L + j) * 3 + 1) * 3;
    const float dx = cx - cj[0], dy = cy - cj[1], dz = cz - cj[2];
    dist[j] = sqrtf(dx * dx + dy * dy + dz * dz) + (j == i ? 1e3f : 0.f);
  }
  __syncthreads();
  const int64_t ii = aa_idx[(int64_t)b * L + i];
  for (int j = threadIdx.x; j < L; j += 256) {
    const float dj = dist[j];
    int rank = 0;
    for (int t = 0; t < L; ++t) {
      const float dt = dist[t];
      rank += (dt < dj || (dt == dj && t < j)) ? 1 : 0;
    }
    if (!(dj == dj)) rank = L;  // NaN distance: every comparison is false -> it would rank first; never a neighbour
    const int64_t jj = aa_idx[(int64_t)b * L + j];
    const int64_t sep = ii > jj ? ii - jj : jj - ii;
    const bool near = (j != i) && sep < kmin;
    mask[(int64_t)bi * L + j] = (rank < k || near) ? 1 : 0;
  }
}

extern "C" int rf_knn_mask(const float* xyz, const int64_t* aa_idx, uint8_t* mask, int B, int L, int k, int kmin,
                           void* stream) {
  if (L > 8192) return RF_EINVAL;
  hipLaunchKernelGGL(knn_mask_kernel, dim3(B * L), dim3(256), L * sizeof(float), (hipStream_t)stream, xyz, aa_idx, mask,
                     L, k < L ? k : L, kmin);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// edge list compaction in row-major (b,i,j) order
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_count_kernel(const uint8_t* mask, int32_t* row_cnt, int L) {
  __shared__ int red[4];
  const int row = blockIdx.x;
  int c = 0;
  for (int j = threadIdx.x; j < L; j += 256) c += mask[(int64_t)row * L + j];
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) row_cnt[row] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void edge_scan_kernel(const int32_t* row_cnt, int32_t* row_off, int32_t* count,
                                                        int rows, int64_t capacity) {
  // single block exclusive scan (rows <= a few thousand)
  __shared__ int part[256];
  const int per = (rows + 255) / 256;
  const int r0 = threadIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  int s = 0;
  for (int r = r0; r < r1; ++r) s += row_cnt[r];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = part[t];
      part[t] = run;
      run += v;
    }
    count[0] = run < capacity ? run : (int32_t)capacity;  // consumers walk [0, count[0]): never past the buffers
    count[1] = run;                                         // true edge count (> capacity: overflow, edges were dropped)
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int r = r0; r < r1; ++r) {
    row_off[r] = run;
    run += row_cnt[r];
  }
}

__global__ __launch_bounds__(256) void edge_write_kernel(const uint8_t* mask, const int32_t* row_off, int32_t* src,
                                                         int32_t* dst, int32_t* eid, int L, int64_t capacity) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int row = blockIdx.x;  // b*L + i
  const int b = row / L;
  if (threadIdx.x == 0) base = row_off[row];
  __syncthreads();
  for (int j0 = 0; j0 < L; j0 += 256) {
    const int j = j0 + threadIdx.x;
    const int m = (j < L) ? mask[(int64_t)row * L + j] : 0;
    const unsigned long long bal = __ballot(m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (j < L) {
      int id = -1;
      if (m) {
        id = off + before;
        if (id < capacity) {
          src[id] = row;
          dst[id] = b * L + j;
        } else {
          id = -1;  // beyond the caller's buffers: dropped (count[1] > capacity reports it)
        }
      }
      eid[(int64_t)row * L + j] = id;
    }
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

extern "C" int rf_edges_from_mask(const uint8_t* mask, int32_t* src, int32_t* dst, int32_t* eid, int32_t* count,
                                  int32_t* row_ws, int B, int L, int64_t capacity, void* stream) {
  if (capacity <= 0 || capacity > 0x7fffffffLL) return RF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int rows = B * L;
  int32_t* row_cnt = row_ws;
  int32_t* row_off = row_ws + rows;
  hipLaunchKernelGGL(edge_count_kernel, dim3(rows), dim3(256), 0, s, mask, row_cnt, L);
  hipLaunchKernelGGL(edge_scan_kernel, dim3(1), dim3(256), 0, s, row_cnt, row_off, count, rows, capacity);
  hipLaunchKernelGGL(edge_write_kernel, dim3(rows), dim3(256), 0, s, mask, row_off, src, dst, eid, L, capacity);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// per-edge geometry: d, r, real SH (closed form), equivariant bases folded with Q_J, radial input
// basis layout per edge (34 floats): [0] b00 | [1..3] b01[a] | [4..6] b10[b] | [7..33] b11[a][b][f]
// ------------------------------------------------------------------------------------------------
#define RF_BASIS_LD 34
__global__ __launch_bounds__(256) void edge_geometry_kernel(const float* xyz, const float* edge_emb, const int32_t* src,
                                                            const int32_t* dst, const int32_t* count, float* basis,
                                                            float* feat, int64_t feat_ld, int L, int d_edge,
                                                            int64_t capacity) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= capacity) return;
  float* bs = basis + e * RF_BASIS_LD;
  float* ft = feat + e * feat_ld;
  if (e >= count[0]) {  // keep the padding rows finite: the radial GEMMs run over the whole capacity
    for (int c = 0; c < RF_BASIS_LD; ++c) bs[c] = 0.f;
    for (int c = 0; c <= d_edge; ++c) ft[c] = 0.f;
    return;
  }
  const int s = src[e], t = dst[e];
  const float* cs = xyz + ((int64_t)s * 3 + 1) * 3;
  const float* ct = xyz + ((int64_t)t * 3 + 1) * 3;
  const float d0 = ct[0] - cs[0], d1 = ct[1] - cs[1], d2 = ct[2] - cs[2];
  const float r = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
  float u0 = 0.f, u1 = 1.f, u2 = 0.f;  // zero vector -> (0,1,0)  (atan2(0,0) = 0 in the reference)
  if (r > 0.f) {
    u0 = d0 / r;
    u1 = d1 / r;
    u2 = d2 / r;
  }
  const float y = u0, z = u1, x = u2;
  const float Y0 = 0.28209479177387814f;
  const float Y1[3] = {-0.4886025119029199f * u0, -0.4886025119029199f * u1, -0.4886025119029199f * u2};
  const float c2 = 1.0925484305920792f;
  const float Y2[5] = {c2 * x * y, c2 * y * z, 0.31539156525252005f * (3.f * z * z - 1.f), c2 * x * z,
                       0.5462742152960396f * (x * x - y * y)};
  const float i3 = 0.5773502691896258f, i6 = 0.4082482904638631f, i10 = 0.31622776601683794f,
              i30 = 0.18257418583505536f;
  bs[0] = Y0;
  for (int a = 0; a < 3; ++a) {
    bs[1 + a] = Y1[a] * i3;
    bs[4 + a] = Y1[a] * i3;
  }
  float K[3][3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      K[a][b][0] = (a == b) ? Y0 * i3 : 0.f;
      K[a][b][1] = 0.f;
      K[a][b][2] = 0.f;
    }
  // J = 1: eps[a,b,c] Y1[c] / sqrt(6)
  K[0][1][1] = Y1[2] * i6;
  K[1][0][1] = -Y1[2] * i6;
  K[1][2][1] = Y1[0] * i6;
  K[2][1][1] = -Y1[0] * i6;
  K[2][0][1] = Y1[1] * i6;
  K[0][2][1] = -Y1[1] * i6;
  // J = 2 (component order: 0 = y, 1 = z, 2 = x)
  K[2][0][2] = K[0][2][2] = -Y2[0] * i10;
  K[0][1][2] = K[1][0][2] = -Y2[1] * i10;
  K[1][1][2] = -2.f * Y2[2] * i30;
  K[2][2][2] = Y2[2] * i30 - Y2[4] * i10;
  K[0][0][2] = Y2[2] * i30 + Y2[4] * i10;
  K[2][1][2] = K[1][2][2] = -Y2[3] * i10;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int f = 0; f < 3; ++f) bs[7 + (a * 3 + b) * 3 + f] = K[a][b][f];
  // radial input [w | r]; w = edge_emb[b, i, j, :] with (src, dst) = (b*L+i, b*L+j)
  const int64_t bb = s / L, i = s % L, j = t % L;
  const float* w = edge_emb + ((bb * L + i) * L + j) * d_edge;
  for (int c = 0; c < d_edge; ++c) ft[c] = w[c];
  ft[d_edge] = r;
}

extern "C" int rf_se3_edge_geometry(const float* xyz, const float* edge_emb, const int32_t* src, const int32_t* dst,
                                    const int32_t* count, float* basis, float* feat, int64_t feat_ld, int L, int d_edge,
                                    int64_t capacity, void* stream) {
  hipLaunchKernelGGL(edge_geometry_kernel, dim3(cdiv(capacity, 256)), dim3(256), 0, (hipStream_t)stream, xyz, edge_emb,
                     src, dst, count, basis, feat, feat_ld, L, d_edge, capacity);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// partial-convolution message: thread per (edge, output channel o)
// msg[e,o,a] = sum_di sum_{i,f} R_di[e,(o*mi+i)*nf+f] * sum_b basis_{di,dout}[a,b,f] * h_di[src,i,b]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se3_message_kernel(const float* R0, const float* R1, const float* basis,
                                                          const float* h0, const float* h1, const int32_t* src,
                                                          const int32_t* count, float* msg, int mo, int dout, int mi0,
                                                          int mi1, int64_t capacity) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e = idx / mo;
  const int o = idx % mo;
  if (e >= capacity || e >= count[0]) return;
  const float* bs = basis + e * RF_BASIS_LD;
  const int s = src[e];
  const int na = 2 * dout + 1;
  float acc[3] = {0.f, 0.f, 0.f};
  if (mi0 > 0) {  // d_in = 0: nf = 1, basis [a][0][0]
    const float* R = R0 + (e * mo + o) * (int64_t)mi0;
    const float* h = h0 + (int64_t)s * mi0;
    float t = 0.f;
    for (int i = 0; i < mi0; ++i) t = fmaf(R[i], h[i], t);
    if (dout == 0)
      acc[0] += t * bs[0];
    else
      for (int a = 0; a < 3; ++a) acc[a] += t * bs[1 + a];
  }
  if (mi1 > 0) {
    const float* h = h1 + (int64_t)s * mi1 * 3;
    if (dout == 0) {  // nf = 1, basis [0][b][0] = bs[4+b]
      const float* R = R1 + (e * mo + o) * (int64_t)mi1;
      float t = 0.f;
      for (int i = 0; i < mi1; ++i) {
        const float hb = bs[4] * h[i * 3] + bs[5] * h[i * 3 + 1] + bs[6] * h[i * 3 + 2];
        t = fmaf(R[i], hb, t);
      }
      acc[0] += t;
    } else {  // nf = 3, basis [a][b][f] = bs[7 + (a*3+b)*3 + f]
      const float* R = R1 + (e * mo + o) * (int64_t)mi1 * 3;
      for (int i = 0; i < mi1; ++i) {
        const float hx = h[i * 3], hy = h[i * 3 + 1], hz = h[i * 3 + 2];
        for (int f = 0; f < 3; ++f) {
          const float rr = R[i * 3 + f];
          for (int a = 0; a < 3; ++a) {
            const float T = bs[7 + (a * 3 + 0) * 3 + f] * hx + bs[7 + (a * 3 + 1) * 3 + f] * hy +
                            bs[7 + (a * 3 + 2) * 3 + f] * hz;
            acc[a] = fmaf(rr, T, acc[a]);
          }
        }
      }
    }
  }
  for (int a = 0; a < na; ++a) msg[(e * mo + o) * na + a] = acc[a];
}

extern "C" int rf_se3_message(const float* R0, const float* R1, const float* basis, const float* h0, const float* h1,
                              const int32_t* src, const int32_t* count, float* msg, int mo, int dout, int mi0, int mi1,
                              int64_t capacity, void* stream) {
  if (dout < 0 || dout > 1) return RF_EINVAL;
  hipLaunchKernelGGL(se3_message_kernel, dim3(cdiv(capacity * mo, 256)), dim3(256), 0, (hipStream_t)stream, R0, R1,
                     basis, h0, h1, src, count, msg, mo, dout, mi0, mi1, capacity);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Fused radial MLP + partial-convolution message (round 4; SURVEY 7.9: "radial MLP -> (x) basis -> matvec in one kernel").
// Reference: RadialFunc (ea/modules.py:246-284: Linear(d_edge+1 -> 32) -> LayerNorm -> ReLU -> Linear(32 -> 32) -> LayerNorm ->
// ReLU -> Linear(32 -> mo*mi*nf)), PairwiseConv's kernel = sum_f R * basis (:287-325), GConvSE3Partial's matvec with the
// source node's features (:612-641).  For output degree `dout` with MO channels
//   msg[e, o, a] = sum_{di} sum_{i, f} R_di[e, (o, i, f)] * T_di[e, i, f, a],
//   R_di[e, :] = radial MLP of net (di, dout) on feat[e, :] = [edge embedding | r],
//   T_di[e, i, f, a] = sum_b basis_{di,dout}[e, a, b, f] h_di[src[e], i, b].
// The round-3 form ran the MLP as ~35 small fp32 GEMM / LayerNorm launches per GSE3Res and wrote every radial output
// R[e, (o, i, f)] (up to 512 floats per net, 15 KB per edge over the 24 nets of a structure-module call) to HBM, to be read
// back by the message kernel: 80 % of the structure module's time (BASELINE.json configs[4]).  Here nothing between feat and
// msg exists in memory: a thread owns TWO edges (packed as float pairs: v_pk_fma_f32, both halves share every weight), the
// net's parameters sit in LDS and are read with wave-uniform addresses (broadcast, conflict-free), the 32-wide hidden
// vectors and their LayerNorms live in registers, the (i, f) loop forms T on the fly from the gathered source features and
// the edge's basis, and the MO x (2 dout + 1) outputs accumulate in registers.  fp32 FMAs throughout (se3_modules.py:164).
// One launch per (value | key, dout).  Packed parameters of one net (host: structure.py GSE3Res._packed), fp32:
//   [W1^T: KI x 32][b1 32][g1 32][e1 32][W2^T: 32 x 32][b2 32][g2 32][e2 32][W3: rows x 32][b3: rows],  rows = MO*mi*nf (o, i, f).
// ------------------------------------------------------------------------------------------------
typedef float rf_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void radial_ln_relu(rf_v2f (&x)[32], const float* g, const float* b, float eps) {
  rf_v2f m = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 32; ++j) m += x[j];
  m *= (1.f / 32.f);
  rf_v2f q = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    x[j] -= m;
    q += x[j] * x[j];
  }
  const rf_v2f rs = {rsqrtf(q.x * (1.f / 32.f) + eps), rsqrtf(q.y * (1.f / 32.f) + eps)};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 gv = ((const float4*)g)[c], bv = ((const float4*)b)[c];
    const float gg[4] = {gv.x, gv.y, gv.z, gv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      rf_v2f y = x[4 * c + t] * rs * gg[t] + bb[t];
      y.x = fmaxf(y.x, 0.f);
      y.y = fmaxf(y.y, 0.f);
      x[4 * c + t] = y;
    }
  }
}

template <int MO, int DOUT>
__global__ __launch_bounds__(256) void se3_radial_message_kernel(const float* __restrict__ feat, int64_t feat_ld, int KI,
                                                                 const float* __restrict__ net0, const float* __restrict__ net1,
                                                                 const float* __restrict__ basis, const float* __restrict__ h0,
                                                                 const float* __restrict__ h1, const int32_t* __restrict__ src,
                                                                 const int32_t* __restrict__ count, float* __restrict__ msg,
                                                                 int mi0, int mi1, float eps, int64_t capacity) {
  constexpr int NA = 2 * DOUT + 1;
  extern __shared__ __attribute__((aligned(16))) float sw[];  // one net's packed parameters
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * 512 + tid, e1 = e0 + 256;
  const int64_t n = count[0] < capacity ? count[0] : capacity;
  if (n <= 0) return;  // (uniform: no edge at all -- the clamped lanes below would read src[0] of an empty list)
  const bool ok0 = e0 < n, ok1 = e1 < n;
  const int64_t ea = ok0 ? e0 : 0, eb = ok1 ? e1 : 0;  // (clamped: lanes without an edge compute on edge 0 and store nothing)
  const int sa = src[ea], sb = src[eb];
  rf_v2f acc[MO][NA];
#pragma unroll
  for (int o = 0; o < MO; ++o)
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[o][a] = (rf_v2f){0.f, 0.f};
  const float* bsa = basis + ea * RF_BASIS_LD;
  const float* bsb = basis + eb * RF_BASIS_LD;
  const float* fa = feat + ea * feat_ld;
  const float* fb = feat + eb * feat_ld;
  const int front = (KI + 3 + 32 + 3) * 32;  // floats before W3

#pragma unroll 1
  for (int di = 0; di < 2; ++di) {
    const int mi = di ? mi1 : mi0;
    if (mi <= 0) continue;  // (uniform)
    const int nf = (di == 1 && DOUT == 1) ? 3 : 1;
    const int rows = MO * mi * nf;
    const float* P = di ? net1 : net0;
    __syncthreads();  // the previous net's parameters are no longer read
    for (int t = tid; t < (front + rows * 33) / 4; t += 256) ((float4*)sw)[t] = ((const float4*)P)[t];
    for (int t = (front + rows * 33) / 4 * 4 + tid; t < front + rows * 33; t += 256) sw[t] = P[t];
    __syncthreads();
    const float* sW1 = sw;                     // [KI][32]
    const float* sb1 = sw + KI * 32;           // b1 | g1 | e1
    const float* sW2 = sb1 + 96;               // [32][32] (k-major)
    const float* sb2 = sW2 + 1024;             // b2 | g2 | e2
    const float* sW3 = sw + front;             // [rows][32]
    const float* sb3 = sW3 + rows * 32;
    rf_v2f hh[32];
    {
      // ---- layer 1: x1[j] = b1[j] + sum_k W1[j][k] feat[k]   (W1 stored k-major: one broadcast row of 32 outputs per k)
      rf_v2f x1[32];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 bv = ((const float4*)sb1)[c];
        x1[4 * c] = (rf_v2f){bv.x, bv.x}; x1[4 * c + 1] = (rf_v2f){bv.y, bv.y};
        x1[4 * c + 2] = (rf_v2f){bv.z, bv.z}; x1[4 * c + 3] = (rf_v2f){bv.w, bv.w};
      }
#pragma unroll 1
      for (int k = 0; k < KI; ++k) {
        const rf_v2f f = {fa[k], fb[k]};
        const float4* w = (const float4*)(sW1 + k * 32);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float4 wv = w[c];
          x1[4 * c] += wv.x * f; x1[4 * c + 1] += wv.y * f; x1[4 * c + 2] += wv.z * f; x1[4 * c + 3] += wv.w * f;
        }
      }
      radial_ln_relu(x1, sb1 + 32, sb1 + 64, eps);
      // ---- layer 2
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 bv = ((const float4*)sb2)[c];
        hh[4 * c] = (rf_v2f){bv.x, bv.x}; hh[4 * c + 1] = (rf_v2f){bv.y, bv.y};
        hh[4 * c + 2] = (rf_v2f){bv.z, bv.z}; hh[4 * c + 3] = (rf_v2f){bv.w, bv.w};
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float4* w = (const float4*)(sW2 + k * 32);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float4 wv = w[c];
          hh[4 * c] += wv.x * x1[k]; hh[4 * c + 1] += wv.y * x1[k]; hh[4 * c + 2] += wv.z * x1[k]; hh[4 * c + 3] += wv.w * x1[k];
        }
      }
      radial_ln_relu(hh, sb2 + 32, sb2 + 64, eps);
    }
    // ---- layer 3 rows (o, i, f) against T[i, f, a], accumulated over (i, f).  The weights of the NEXT row are read (8 broadcast
    // ds_read_b128 + its bias) before the 32 dependent FMAs of the current one, across the (f, o) jobs of an input channel and
    // across input channels: hipcc's own order read them two at a time right in front of their uses and exposed the LDS latency
    // every four FMAs (two waves per SIMD do not hide it)
    float4 wb[2][8];
    float bq[2];
    auto rd_row = [&](int row, float4 (&w)[8], float& bias) {
      const float4* p = (const float4*)(sW3 + row * 32);
#pragma unroll
      for (int c = 0; c < 8; ++c) w[c] = p[c];
      bias = sb3[row];
    };
    // (measured, BASELINE configs[4]: the read-ahead pays where a channel has many rows -- MO = 32: 548 -> 491 us -- and costs
    // registers and copies where it has four: 167 -> 206 us; those instances keep the compiler's order)
    constexpr bool PIPE = MO >= 16;
    if constexpr (PIPE) rd_row(0, wb[0], bq[0]);  // job (i = 0, f = 0, o = 0)
#pragma unroll 1
    for (int i = 0; i < mi; ++i) {
      rf_v2f T[3][NA];
      if (di == 0) {
        const rf_v2f x = {h0[(int64_t)sa * mi0 + i], h0[(int64_t)sb * mi0 + i]};
#pragma unroll
        for (int a = 0; a < NA; ++a) T[0][a] = x * (rf_v2f){DOUT == 0 ? bsa[0] : bsa[1 + a], DOUT == 0 ? bsb[0] : bsb[1 + a]};
      } else {
        const float* pa = h1 + ((int64_t)sa * mi1 + i) * 3;
        const float* pb = h1 + ((int64_t)sb * mi1 + i) * 3;
        const rf_v2f hx = {pa[0], pb[0]}, hy = {pa[1], pb[1]}, hz = {pa[2], pb[2]};
        if (DOUT == 0) {
          T[0][0] = (rf_v2f){bsa[4], bsb[4]} * hx + (rf_v2f){bsa[5], bsb[5]} * hy + (rf_v2f){bsa[6], bsb[6]} * hz;
        } else {
#pragma unroll
          for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int a = 0; a < NA; ++a)
              T[f][a] = (rf_v2f){bsa[7 + (a * 3 + 0) * 3 + f], bsb[7 + (a * 3 + 0) * 3 + f]} * hx +
                        (rf_v2f){bsa[7 + (a * 3 + 1) * 3 + f], bsb[7 + (a * 3 + 1) * 3 + f]} * hy +
                        (rf_v2f){bsa[7 + (a * 3 + 2) * 3 + f], bsb[7 + (a * 3 + 2) * 3 + f]} * hz;
        }
      }
      constexpr int NFMAX = (DOUT == 1) ? 3 : 1;   // nf of the di = 1 net (di = 0: 1); jobs j = f * MO + o, f < nf
#pragma unroll
      for (int j = 0; j < NFMAX * MO; ++j) {
        const int f = j / MO, o = j % MO;
        if (f >= nf) break;
        // next job: (f, o + 1), (f + 1, 0) or (i + 1: f = 0, o = 0) -- one row past the net's last one at the very end: still inside
        // this net's LDS image (its bias block follows), read and never used
        const bool last = (j + 1 == nf * MO);
        const int nrow = last ? (i + 1) * nf : (((j + 1) % MO) * mi + i) * nf + (j + 1) / MO;
        if constexpr (PIPE) {
          rd_row(nrow, wb[(j + 1) & 1], bq[(j + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          rd_row((o * mi + i) * nf + f, wb[j & 1], bq[j & 1]);
        }
        const float4 (&w)[8] = wb[j & 1];
        rf_v2f r = {bq[j & 1], bq[j & 1]};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          r += w[c].x * hh[4 * c]; r += w[c].y * hh[4 * c + 1]; r += w[c].z * hh[4 * c + 2]; r += w[c].w * hh[4 * c + 3];
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[o][a] += r * T[f][a];
        if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
      }
      if (PIPE && ((nf * MO) & 1)) {  // odd job count: the prefetched row sits in buffer 1, the next channel starts from buffer 0
#pragma unroll
        for (int c = 0; c < 8; ++c) wb[0][c] = wb[1][c];
        bq[0] = bq[1];
      }
    }
  }
  if (ok0) {
    float* m = msg + e0 * (MO * NA);
#pragma unroll
    for (int o = 0; o < MO; ++o)
#pragma unroll
      for (int a = 0; a < NA; ++a) m[o * NA + a] = acc[o][a].x;
  }
  if (ok1) {
    float* m = msg + e1 * (MO * NA);
#pragma unroll
    for (int o = 0; o < MO; ++o)
#pragma unroll
      for (int a = 0; a < NA; ++a) m[o * NA + a] = acc[o][a].y;
  }
}

static inline int64_t radial_net_floats(int KI, int64_t rows) { return (int64_t)(KI + 3 + 32 + 3) * 32 + rows * 33; }

template <int MO, int DOUT>
static int launch_radial_message(const float* feat, int64_t feat_ld, int KI, const float* net0, const float* net1,
                                 const float* basis, const float* h0, const float* h1, const int32_t* src, const int32_t* count,
                                 float* msg, int mi0, int mi1, float eps, int64_t capacity, hipStream_t s) {
  const int nf1 = DOUT == 1 ? 3 : 1;
  const int rows = MO * (mi0 > mi1 * nf1 ? mi0 : mi1 * nf1);
  const size_t lds = (size_t)radial_net_floats(KI, rows) * sizeof(float);
  if (lds > 160 * 1024) return RF_EINVAL;
  if (lds > 64 * 1024)
    if (const int e = rf_enable_big_lds<se3_radial_message_kernel<MO, DOUT>>()) return e;
  hipLaunchKernelGGL((se3_radial_message_kernel<MO, DOUT>), dim3(cdiv(capacity, 512)), dim3(256), lds, s, feat, feat_ld, KI, net0,
                     net1, basis, h0, h1, src, count, msg, mi0, mi1, eps, capacity);
  return rf_launch_status();
}

// 1 when (mo, dout, mi0, mi1, d_edge + 1) has a fused instance (rf_se3_radial_message would launch), 0 otherwise
extern "C" int rf_se3_radial_message_supported(int mo, int dout, int mi0, int mi1, int ki) {
  if (dout < 0 || dout > 1 || mi0 < 0 || mi1 < 0 || (mi0 == 0 && mi1 == 0) || ki < 1 || ki > 4096) return 0;
  // instances: degree-1 outputs have 3 (layer 4: the displacement channels) or 4 (num_channels / div) channels on the forward
  // path (rf.py:774-784); degree-0 outputs 4 or d_state
  if (dout == 1 ? !(mo == 3 || mo == 4) : !(mo == 4 || mo == 8 || mo == 16 || mo == 32)) return 0;
  const int nf1 = dout == 1 ? 3 : 1;
  const int64_t rows = (int64_t)mo * (mi0 > mi1 * nf1 ? mi0 : mi1 * nf1);
  return radial_net_floats(ki, rows) * 4 <= 160 * 1024 ? 1 : 0;
}

extern "C" int rf_se3_radial_message(const float* feat, int64_t feat_ld, int ki, const float* net0, const float* net1,
                                     const float* basis, const float* h0, const float* h1, const int32_t* src,
                                     const int32_t* count, float* msg, int mo, int dout, int mi0, int mi1, float ln_eps,
                                     int64_t capacity, void* stream) {
  if (!rf_se3_radial_message_supported(mo, dout, mi0, mi1, ki)) return RF_EINVAL;
  if (!feat || !basis || !src || !count || !msg || capacity <= 0 || feat_ld < ki || (mi0 > 0 && (!net0 || !h0)) ||
      (mi1 > 0 && (!net1 || !h1)))
    return RF_EINVAL;
  if (((uintptr_t)net0 % 16) || ((uintptr_t)net1 % 16)) return RF_EALIGN;
  hipStream_t s = (hipStream_t)stream;
#define RF_RM(MO_, DO_)                                                                                                  \
  if (mo == MO_ && dout == DO_)                                                                                          \
    return launch_radial_message<MO_, DO_>(feat, feat_ld, ki, net0, net1, basis, h0, h1, src, count, msg, mi0, mi1, ln_eps, \
                                           capacity, s);
  RF_RM(3, 1) RF_RM(4, 1) RF_RM(4, 0) RF_RM(8, 0) RF_RM(16, 0) RF_RM(32, 0)
#undef RF_RM
  return RF_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// graph attention: one wave per (dst node, head); incoming edges found through the dense eid map
// ------------------------------------------------------------------------------------------------
#define RF_ATT_MAXPL 16  // candidates per lane: L <= 1024
__global__ __launch_bounds__(256) void se3_attention_kernel(const float* k0, const float* k1, const float* q0,
                                                            const float* q1, const float* v0, const float* v1,
                                                            const int32_t* eid, float* out0, float* out1, int heads,
                                                            int mk0, int mk1, int mv0, int mv1, int V, int L,
                                                            float inv_sqrt_nfeat, int64_t o0_ld, int64_t o1_ld) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)V * heads) return;
  const int node = wid / heads, hd = wid % heads;
  const int b = node / L, j = node % L;
  const int ck0 = mk0 / heads, ck1 = mk1 / heads, cv0 = mv0 / heads, cv1 = mv1 / heads;
  float lg[RF_ATT_MAXPL];
  int ids[RF_ATT_MAXPL];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < RF_ATT_MAXPL; ++t) {
    const int i = lane + 64 * t;
    int id = -1;
    float a = -INFINITY;
    if (i < L) id = eid[((int64_t)b * L + i) * L + j];
    if (id >= 0) {
      a = 0.f;
      for (int c = 0; c < ck0; ++c) a = fmaf(k0[(int64_t)id * mk0 + hd * ck0 + c], q0[(int64_t)node * mk0 + hd * ck0 + c], a);
      for (int c = 0; c < ck1 * 3; ++c)
        a = fmaf(k1[((int64_t)id * mk1 + hd * ck1) * 3 + c], q1[((int64_t)node * mk1 + hd * ck1) * 3 + c], a);
      a *= inv_sqrt_nfeat;
    }
    lg[t] = a;
    ids[t] = id;
    mx = fmaxf(mx, a);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < RF_ATT_MAXPL; ++t) {
    const float p = ids[t] >= 0 ? __expf(lg[t] - mx) : 0.f;
    lg[t] = p;
    s += p;
  }
  s = wave_sum(s);
  const float inv = s > 0.f ? 1.f / s : 0.f;  // in-degree 0 -> zeros (fn.sum over no messages)
  for (int c = 0; c < cv0; ++c) {
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < RF_ATT_MAXPL; ++t)
      if (ids[t] >= 0) a = fmaf(lg[t], v0[(int64_t)ids[t] * mv0 + hd * cv0 + c], a);
    a = wave_sum(a);
    if (lane == 0) out0[(int64_t)node * o0_ld + hd * cv0 + c] = a * inv;
  }
  for (int c = 0; c < cv1 * 3; ++c) {
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < RF_ATT_MAXPL; ++t)
      if (ids[t] >= 0) a = fmaf(lg[t], v1[((int64_t)ids[t] * mv1 + hd * cv1) * 3 + c], a);
    a = wave_sum(a);
    if (lane == 0) out1[(int64_t)node * o1_ld + hd * cv1 * 3 + c] = a * inv;
  }
}

extern "C" int rf_se3_attention(const float* k0, const float* k1, const float* q0, const float* q1, const float* v0,
                                const float* v1, const int32_t* eid, float* out0, float* out1, int heads, int mk0,
                                int mk1, int mv0, int mv1, int V, int L, int64_t out0_ld, int64_t out1_ld, void* stream) {
  if (L > 64 * RF_ATT_MAXPL) return RF_EINVAL;
  if (out0_ld <= 0) out0_ld = mv0;      // elements per node row of out0 / out1 (> mv0 / 3*mv1: the attention writes the
  if (out1_ld <= 0) out1_ld = 3 * mv1;  // leading channels of the GCat buffer, ea/modules.py:903-928)
  if (out0_ld < mv0 || out1_ld < 3 * mv1) return RF_EINVAL;
  const float nfeat = (float)(mk0 + 3 * mk1);
  hipLaunchKernelGGL(se3_attention_kernel, dim3(cdiv((int64_t)V * heads, 4)), dim3(256), 0, (hipStream_t)stream, k0, k1,
                     q0, q1, v0, v1, eid, out0, out1, heads, mk0, mk1, mv0, mv1, V, L, 1.0f / sqrtf(nfeat), out0_ld, out1_ld);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// GNormBias, GAttentiveSelfInt pieces, coordinate update
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_bias_kernel(const float* v, const float* bias, float* y, int64_t VM, int m,
                                                        int nc) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (node, channel)
  if (e >= VM) return;
  const int c = e % m;
  float n2 = 0.f;
  for (int a = 0; a < nc; ++a) n2 = fmaf(v[e * nc + a], v[e * nc + a], n2);
  const float norm = fmaxf(sqrtf(n2), 1e-12f);
  const float t = fmaxf(norm + bias[c], 0.f);
  for (int a = 0; a < nc; ++a) y[e * nc + a] = t * (v[e * nc + a] / norm);
}

extern "C" int rf_se3_norm_bias(const float* v, const float* bias, float* y, int64_t V, int m, int deg, void* stream) {
  hipLaunchKernelGGL(norm_bias_kernel, dim3(cdiv(V * m, 256)), dim3(256), 0, (hipStream_t)stream, v, bias, y, V * m, m,
                     2 * deg + 1);
  return rf_launch_status();
}

__global__ __launch_bounds__(256) void gram_kernel(const float* v, float* s, int64_t total, int m, int nc) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (node, a, b)
  if (e >= total) return;
  const int b = e % m, a = (e / m) % m;
  const int64_t n = e / ((int64_t)m * m);
  float d = 0.f;
  for (int c = 0; c < nc; ++c) d = fmaf(v[(n * m + a) * nc + c], v[(n * m + b) * nc + c], d);
  const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  s[e] = fmaxf(fabsf(d), 1e-12f) * sg;
}

extern "C" int rf_se3_gram(const float* v, float* s, int64_t V, int m, int deg, void* stream) {
  const int64_t total = V * m * m;
  hipLaunchKernelGGL(gram_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, v, s, total, m,
                     2 * deg + 1);
  return rf_launch_status();
}

// y[n,o,c] = sum_m softmax_m(att[n,o,:])[m] * x[n,m,c]; one thread per (n,o)
__global__ __launch_bounds__(256) void attn_apply_kernel(const float* att, const float* x, float* y, int64_t total,
                                                         int m_out, int m_in, int nc) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t n = e / m_out;
  const float* a = att + e * m_in;
  float mx = -INFINITY;
  for (int m = 0; m < m_in; ++m) mx = fmaxf(mx, a[m]);
  float s = 0.f, acc[3] = {0.f, 0.f, 0.f};
  for (int m = 0; m < m_in; ++m) {
    const float p = __expf(a[m] - mx);
    s += p;
    for (int c = 0; c < nc; ++c) acc[c] = fmaf(p, x[(n * m_in + m) * nc + c], acc[c]);
  }
  for (int c = 0; c < nc; ++c) y[e * nc + c] = acc[c] / s;
}

extern "C" int rf_se3_attn_apply(const float* att, const float* x, float* y, int64_t V, int m_out, int m_in, int deg,
                                 void* stream) {
  const int64_t total = V * m_out;
  hipLaunchKernelGGL(attn_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, att, x, y, total,
                     m_out, m_in, 2 * deg + 1);
  return rf_launch_status();
}

__global__ __launch_bounds__(256) void coord_apply_kernel(const float* xyz, const float* disp, float* out, int64_t nres) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (res, comp)
  if (e >= nres * 3) return;
  const int64_t r = e / 3;
  const int c = e % 3;
  const float ca = xyz[(r * 3 + 1) * 3 + c] + disp[(r * 3 + 1) * 3 + c];
  out[(r * 3 + 0) * 3 + c] = ca + disp[(r * 3 + 0) * 3 + c];
  out[(r * 3 + 1) * 3 + c] = ca;
  out[(r * 3 + 2) * 3 + c] = ca + disp[(r * 3 + 2) * 3 + c];
}

extern "C" int rf_coord_apply(const float* xyz, const float* disp, float* xyz_out, int64_t nres, void* stream) {
  hipLaunchKernelGGL(coord_apply_kernel, dim3(cdiv(nres * 3, 256)), dim3(256), 0, (hipStream_t)stream, xyz, disp,
                     xyz_out, nres);
  return rf_launch_status();
}

// type-1 input features (rf.py:807): y[r,a,:] = xyz[r,a,:] - xyz[r,CA,:]
__global__ __launch_bounds__(256) void center_ca_kernel(const float* xyz, float* y, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int64_t r = e / 9;
  const int c = e % 3;
  y[e] = xyz[e] - xyz[(r * 3 + 1) * 3 + c];
}

extern "C" int rf_center_ca(const float* xyz, float* y, int64_t nres, void* stream) {
  hipLaunchKernelGGL(center_ca_kernel, dim3(cdiv(nres * 9, 256)), dim3(256), 0, (hipStream_t)stream, xyz, y, nres * 9);
  return rf_launch_status();
}
