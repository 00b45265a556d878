// Convolutions with a thin input (Cin <= 4) and 32 output channels on the fp32 matrix cores:
//   EdgeAwareRefinement.conv2d_feature = nn.Conv2d(4, 32, 3, padding=1) over cat([disparity, rgb])
//     (adaptive_stereo/models/stereo_net.py:89-94, 116-118)
//   FeatureExtractorNetwork.downsample[0] = nn.Conv2d(3, 32, 5, stride=2, padding=2)  (:61-69)
//
// Input layout "PCL4": float buf[B][H+2*ph][W+2*pw][4] with a zero halo — one 16-byte pixel.
// GEMM view: Z[pix][co] = sum_{tap} sum_{c<4} X4[pix*stride + off(tap)][c] * W[co][c][tap],
// K = 4*taps (36 or 100).  Per tap a lane loads its pixel's float4 once and issues two
// 32x32x2 MFMAs (k = lane>>5 picks channel 2j+h).  These layers are HBM-bound (16 B in, 128 B out
// per pixel); the matrix core is used because the epilogue (BatchNorm partials, fused affine) and
// the output tile layout are then shared with conv32.
#include "as_common.h"
#include "conv_epilogue.h"

struct Conv4Args {
  const float* x4;
  const float* wp;        // [tap][j][h][co]
  EpilogueArgs ep;
  PclDev gin, gout;       // gin describes the PCL4 tensor (4 floats per pixel)
  int M, stride, ntaps;
  int tap_off[AS_MAX_TAPS];
};

__global__ __launch_bounds__(256) void conv4_fwd_kernel(Conv4Args p) {
  __shared__ float red[4][32];
  __shared__ float bmean[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int v = (blockIdx.x * 4 + wave) * 32 + li;
  const bool valid = v < p.M;
  int in_vox, out_vox;
  ConvMap m; m.Hl = p.gout.H; m.Wl = p.gout.W; m.in_stride = p.stride; m.out_stride = 1; m.out_oy = 0; m.out_ox = 0;
  conv_decode(valid ? v : p.M - 1, p.gin, p.gout, m, in_vox, out_vox);
  const float* xa = p.x4 + (long)in_vox * 4;
  f32x16 acc;
  conv_init_acc(acc, p.ep.bias, li);
  for (int tp = 0; tp < p.ntaps; ++tp) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(xa + (long)p.tap_off[tp] * 4);
    const float b0 = p.wp[(tp * 2 + 0) * 64 + lane];
    const float b1 = p.wp[(tp * 2 + 1) * 64 + lane];
    const float a0 = h ? q.y : q.x;
    const float a1 = h ? q.w : q.z;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
  }
  TileStats ts;
  conv_epilogue(acc, p.ep, out_vox, valid, min(128, p.M - (int)blockIdx.x * 128), red, bmean, &ts);
  stats_write(p.ep, blockIdx.x, ts);
}

// packed[t][j][h][co] = w[co][c = 2j+h][t]  (0 for c >= Cin)
__global__ void conv4_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int T, int Cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * 128) return;
  const int co = idx & 31, h = (idx >> 5) & 1, j = (idx >> 6) & 1, t = idx >> 7;
  const int c = 2 * j + h;
  packed[idx] = c < Cin ? w[((long)co * Cin + c) * T + t] : 0.f;
}

// x4[b][y][x][:] = (ch0[b,0,y,x] if given), img[b,0..C-1,y,x], zero-filled to 4 channels.
__global__ __launch_bounds__(256) void pack_in4_kernel(const float* __restrict__ ch0, const float* __restrict__ img, int C,
                                                        float* __restrict__ x4, PclDev g) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long plane = (long)g.H * g.W;
  if (i >= (long)g.B * plane) return;
  const int x = i % g.W, y = (i / g.W) % g.H, b = i / plane;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  int n = 0;
  if (ch0) v[n++] = ch0[i];
  for (int c = 0; c < C && n < 4; ++c) v[n++] = img[((long)b * C + c) * plane + (long)y * g.W + x];
  *reinterpret_cast<f32x4*>(x4 + g.vox(b, 0, y, x) * 4) = (f32x4){v[0], v[1], v[2], v[3]};
}

// ---- weight gradient: dW[co][c][t] = sum_pix X4[pix*stride + off(t)][c] * G[pix][co] ------------------------
// MFMA rows i = k = 4*t + c in blocks of 32 (NB blocks cover 4*T), columns j = co, reduction over pixels.
struct Wgrad4Args {
  const float* x4;
  const float* gz;
  float* partial;      // [nchunks][NB][32][32]
  float* partial_db;   // [nchunks][32]
  PclDev gin, gout;
  int rows, rows_per_chunk, ntaps, stride;
  int tap_off[AS_MAX_TAPS];
};

template <int NB>
__global__ __launch_bounds__(256) void conv4_wgrad_kernel(Wgrad4Args p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [3][NB*16][64] + [4][32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int chunk = blockIdx.x;
  const int W = p.gout.W, H = p.gout.H;
  const int r0 = chunk * p.rows_per_chunk;
  const int r1 = min(p.rows, r0 + p.rows_per_chunk);

  f32x16 acc[NB];
  int koff[NB];          // float offset of this lane's (tap, channel) relative to the pixel anchor
  bool kok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int k = nb * 32 + li;
    kok[nb] = k < 4 * p.ntaps;
    const int t = kok[nb] ? (k >> 2) : 0;
    koff[nb] = p.tap_off[t] * 4 + (k & 3);
  }
  float bsum = 0.f;
  // loads of 8 steps are issued as a group before their MFMAs (see conv32_wgrad_kernel)
  constexpr int U = 8;
  const int nsteps = (W + 1) >> 1;
  for (int row = r0 + wave; row < r1; row += 4) {
    const int y = row % H, b = row / H;
    const float* xr = p.x4 + p.gin.vox(b, 0, y * p.stride, 0) * 4;
    const float* gr = p.gz + p.gout.vox(b, 0, y, 0) * 32 + li;
    for (int s0 = 0; s0 < nsteps; s0 += U) {
      float bv[U], av[U][NB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int xc = 2 * (s0 + u) + h;
        const bool ok = xc < W;
        const int xcl = ok ? xc : W - 1;
        const float g0 = gr[xcl * 32];
        bv[u] = ok ? g0 : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float a0 = xr[xcl * p.stride * 4 + koff[nb]];
          av[u][nb] = kok[nb] ? a0 : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bsum += bv[u];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][nb], bv[u], acc[nb], 0, 0, 0);
      }
    }
  }
  float* slab = lds;
  float* dbs = lds + 3 * NB * 16 * 64;
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) dbs[wave * 32 + li] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[((wave - 1) * NB * 16 + nb * 16 + r) * 64 + lane] = acc[nb][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float* out = p.partial + ((long)chunk * NB + nb) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nb][r];
        v += slab[(0 * NB * 16 + nb * 16 + r) * 64 + lane];
        v += slab[(1 * NB * 16 + nb * 16 + r) * 64 + lane];
        v += slab[(2 * NB * 16 + nb * 16 + r) * 64 + lane];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[i * 32 + li] = v;
      }
    }
    if (h == 0) p.partial_db[chunk * 32 + li] = dbs[li] + dbs[32 + li] + dbs[64 + li] + dbs[96 + li];
  }
}

// dW[co][c][t] = sum_chunks partial[chunk][k>>5][k&31][co], k = 4t + c
// 32 lanes per output element: lane q sums chunks q, q+32, ..., then a fixed-order butterfly (deterministic).
__global__ void conv4_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_db,
                                          int nchunks, int NB, int T, int Cin, float* __restrict__ dW,
                                          float* __restrict__ db, int accumulate) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int q = threadIdx.x & 31;
  const int total = 32 * Cin * T;
  float s = 0.f;
  if (idx < total) {
    const int t = idx % T, c = (idx / T) % Cin, co = idx / (T * Cin);
    const int k = 4 * t + c;
    for (int ch = q; ch < nchunks; ch += 32) s += partial[(((long)ch * NB + (k >> 5)) * 32 + (k & 31)) * 32 + co];
  } else if (db != nullptr && idx < total + 32) {
    for (int ch = q; ch < nchunks; ch += 32) s += partial_db[ch * 32 + (idx - total)];
  }
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, 32);
  if (q != 0) return;
  if (idx < total) dW[idx] = accumulate ? dW[idx] + s : s;
  else if (db != nullptr && idx < total + 32) db[idx - total] = accumulate ? db[idx - total] + s : s;
}

// ---- host -------------------------------------------------------------------------------------------------
static int fill_taps4(const as_pcl* gin, const as_conv_shape* s, int* tap_off) {
  const int Wp = gin->W + 2 * gin->pw;
  int n = 0;
  for (int j = 0; j < s->kh; ++j)
    for (int l = 0; l < s->kw; ++l) tap_off[n++] = (j * s->dil - s->pad_h) * Wp + (l * s->dil - s->pad_w);
  return n;
}

static int check4(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s, const char* who) {
  AS_CHECK_ARG(gin && gout && s && as_pcl_ok(gout), "%s: bad geometry", who);
  AS_CHECK_ARG(gin->D == 1 && gout->D == 1 && gin->pd == 0 && gout->pd == 0 && s->kd == 1, "%s: 2-D only", who);
  AS_CHECK_ARG(gin->B == gout->B && gin->B > 0 && gin->H > 0 && gin->W > 0, "%s: bad input extent", who);
  const int T = s->kh * s->kw;
  AS_CHECK_ARG(T >= 1 && T <= AS_MAX_TAPS && s->dil >= 1 && s->stride >= 1, "%s: unsupported kernel", who);
  const int eh = (gin->H + 2 * s->pad_h - s->dil * (s->kh - 1) - 1) / s->stride + 1;
  const int ew = (gin->W + 2 * s->pad_w - s->dil * (s->kw - 1) - 1) / s->stride + 1;
  AS_CHECK_ARG(eh == gout->H && ew == gout->W, "%s: output extent mismatch", who);
  const int hi_h = (gout->H - 1) * s->stride + s->dil * (s->kh - 1) - s->pad_h - (gin->H - 1);
  const int hi_w = (gout->W - 1) * s->stride + s->dil * (s->kw - 1) - s->pad_w - (gin->W - 1);
  AS_CHECK_ARG(s->pad_h <= gin->ph && s->pad_w <= gin->pw && hi_h <= gin->ph && hi_w <= gin->pw,
               "%s: input halo too small", who);
  return AS_OK;
}

extern "C" int64_t as_pcl4_numel(const as_pcl* g) {
  if (!g) return -1;
  return (int64_t)g->B * (g->H + 2 * g->ph) * (g->W + 2 * g->pw) * 4;
}

extern "C" int as_pack_in4(const float* ch0, const float* img, int C, float* x4, const as_pcl* g, void* stream) {
  AS_CHECK_ARG(g && g->D == 1 && g->pd == 0 && g->B > 0 && g->H > 0 && g->W > 0, "as_pack_in4: bad geometry");
  AS_CHECK_ARG(img && x4 && C >= 1 && C + (ch0 ? 1 : 0) <= 4, "as_pack_in4: bad argument");
  const long n = (long)g->B * g->H * g->W;
  hipLaunchKernelGGL(pack_in4_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, ch0, img, C, x4,
                     as_make_dev(g));
  AS_CHECK_LAUNCH("as_pack_in4");
  return AS_OK;
}

extern "C" int as_conv4_pack_weights(const float* w, int Cin, float* packed, const as_conv_shape* s, void* stream) {
  AS_CHECK_ARG(w && packed && s && Cin >= 1 && Cin <= 4, "as_conv4_pack_weights: bad argument");
  const int T = s->kh * s->kw;
  AS_CHECK_ARG(T >= 1 && T <= AS_MAX_TAPS, "as_conv4_pack_weights: %d taps unsupported", T);
  hipLaunchKernelGGL(conv4_pack_kernel, dim3(as_div_up(T * 128, 256)), dim3(256), 0, (hipStream_t)stream, w, packed, T, Cin);
  AS_CHECK_LAUNCH("as_conv4_pack_weights");
  return AS_OK;
}

extern "C" int as_conv4_fwd(const float* x4, const as_pcl* gin, const float* packed_w, const float* bias,
                            float* z, const as_pcl* gout, const as_conv_shape* s,
                            int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                            float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  if (int e = check4(gin, gout, s, "as_conv4_fwd")) return e;
  AS_CHECK_ARG(x4 && packed_w && z, "as_conv4_fwd: null pointer");
  AS_CHECK_ARG(epilogue_args_ok(epilogue, ep_scale, ep_shift, stat_mean, stat_m2, stat_cnt), "as_conv4_fwd: bad epilogue arguments");
  Conv4Args a;
  a.x4 = x4; a.wp = packed_w;
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = ep_scale; a.ep.ep_shift = ep_shift; a.ep.residual = nullptr;
  a.ep.stat_mean = epilogue == 0 ? stat_mean : nullptr; a.ep.stat_m2 = epilogue == 0 ? stat_m2 : nullptr;
  a.ep.stat_cnt = epilogue == 0 ? stat_cnt : nullptr;
  a.ep.epilogue = epilogue; a.ep.slope = slope;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  const int64_t M = (int64_t)gout->B * gout->H * gout->W;
  a.M = (int)M; a.stride = s->stride; a.ntaps = fill_taps4(gin, s, a.tap_off);
  hipLaunchKernelGGL(conv4_fwd_kernel, dim3(as_div_up(M, 128)), dim3(256), 0, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_conv4_fwd");
  return AS_OK;
}

static void plan4(const as_pcl* gout, const as_conv_shape* s, int* nb, int* rpc, int* nchunks) {
  const int T = s->kh * s->kw;
  *nb = (4 * T + 31) / 32;
  const int rows = gout->B * gout->H;
  int want = 1024;
  int r = (rows + want - 1) / want;
  if (r < 4) r = 4;
  *rpc = r;
  *nchunks = (rows + r - 1) / r;
}

extern "C" int64_t as_conv4_wgrad_workspace(const as_pcl* gout, const as_conv_shape* s) {
  if (!gout || !s || !as_pcl_ok(gout)) return -1;
  int nb, rpc, nchunks;
  plan4(gout, s, &nb, &rpc, &nchunks);
  return (int64_t)nchunks * nb * 1024 + (int64_t)nchunks * 32;
}

template <int NB>
static void launch4(const Wgrad4Args& a, int nchunks, hipStream_t st) {
  const size_t lds = (size_t)(3 * NB * 16 * 64 + 4 * 32) * sizeof(float);
  hipLaunchKernelGGL(conv4_wgrad_kernel<NB>, dim3(nchunks), dim3(256), lds, st, a);
}

extern "C" int as_conv4_wgrad(const float* x4, const as_pcl* gin, const float* gz, const as_pcl* gout,
                              const as_conv_shape* s, int Cin, float* dW, float* db, int accumulate, float* workspace,
                              void* stream) {
  if (int e = check4(gin, gout, s, "as_conv4_wgrad")) return e;
  AS_CHECK_ARG(x4 && gz && dW && workspace && Cin >= 1 && Cin <= 4, "as_conv4_wgrad: bad argument");
  int nb, rpc, nchunks;
  plan4(gout, s, &nb, &rpc, &nchunks);
  AS_CHECK_ARG(nb >= 1 && nb <= 4, "as_conv4_wgrad: kernel too large");
  Wgrad4Args a;
  a.x4 = x4; a.gz = gz; a.partial = workspace; a.partial_db = workspace + (int64_t)nchunks * nb * 1024;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  a.rows = gout->B * gout->H; a.rows_per_chunk = rpc; a.stride = s->stride;
  a.ntaps = fill_taps4(gin, s, a.tap_off);
  hipStream_t st = (hipStream_t)stream;
  switch (nb) {
    case 1: launch4<1>(a, nchunks, st); break;
    case 2: launch4<2>(a, nchunks, st); break;
    case 3: launch4<3>(a, nchunks, st); break;
    default: launch4<4>(a, nchunks, st); break;
  }
  AS_CHECK_LAUNCH("as_conv4_wgrad");
  const int T = a.ntaps;
  hipLaunchKernelGGL(conv4_wgrad_reduce_kernel, dim3(as_div_up((32 * Cin * T + 32) * 32, 256)), dim3(256), 0, st, a.partial,
                     a.partial_db, nchunks, nb, T, Cin, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv4_wgrad(reduce)");
  return AS_OK;
}
