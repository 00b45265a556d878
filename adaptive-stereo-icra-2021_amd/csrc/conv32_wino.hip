// Training forward of a full-resolution refinement layer (3x3, dilation 1/2/4/8, stride 1, 32->32: stereo_net.py:10-18,
// 33-51, 97) with the previous layer's BatchNorm + LeakyReLU (+ skip) applied to its operand on the way in — the job of
// conv32_act.hip — computed with the minimal-filtering algorithm F(2x2, 3x3) (Winograd; Lavin & Gray 2016):
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A     per 2x2 output tile, 4x4 input tile d, 3x3 filter g,
// 16 multiplications per 2x2 outputs and channel pair instead of 36: 4 x 32x32 matrix products per output pixel where the
// direct form needs 9.  A dilated layer is d*d interleaved dense layers: the tile of dilation d is the pixels
// (y, y+d) x (x, x+d) and its input tile the rows y-d..y+2d, columns x-d..x+2d in steps of d.
//
// A workgroup walks a comb of rows (y = r, r+d, ..) of a 64-pixel column segment, two comb rows per step, over a ring of four
// activated rows in LDS (conv32_act.hip's scheme with a pair of rows where that has one).  The 32 tiles of a step are the 32
// rows of the matrix instruction; the four waves split the sixteen products by the ROW r of the transformed tile:
//   wave r:  R = (B^T d)[r] — a signed sum of two input rows — for the four tile columns, V[r][c] = (R B)[c]: both lane-local
//            (the lane IS the tile; its 16 channels are the K half it feeds), 32 vector operations per 16 MFMAs;
//            M[r][c] += V[r][c] x U[r][c]: 4 x 16 MFMAs, the wave's four U matrices resident in 64 registers;
//            T[j] = (M[r] A)[j]: two 16-register tiles, written to LDS
//   B1
//   wave w = (i, j): Y[i][j] = (A^T T)[i][j] = a signed sum of three waves' T[j]; + bias; 16 stores; BatchNorm moments
//   rows j+3, j+4: activate, by-product, into the two freed slots
//   B2
// Two workgroups per CU (74,496 B of LDS each): one's vector phases under the other's matrix phase.
// The LDS rows are swizzled for the stride-2d tile gather: with key(v) = the voxel index with bit log2(d) removed, the 32
// tiles of one read are 32 consecutive keys; key bit 0 selects the bank half (it is the voxel's parity for d > 1; for d = 1
// voxels are stored with bits 0 and 1 swapped) and key bits 1-3 the 16-byte slot.
//
// MODE 2 is the DATA GRADIENT of such a layer (the first half of its backward pass; conv32_wino_wgrad.hip is the second):
// the staged operand is g_z = stage 3 of the layer's BatchNorm backward applied to (g_a, z) on the way in — written back once
// as a by-product for the weight gradient —, the filter is the transposed one (AS_PACK_WINO_T), the epilogue adds the skip
// connection (g_x = dgrad(g_z) + g_a) and forms stage 1 of the NEXT BatchNorm backward from the g_x register tile: the
// data-gradient half of conv32_bwd.hip with 4 matrix products per pixel instead of 9.
//
// MODE 3 is the INFERENCE block: the operand is staged as is, the epilogue applies the folded BatchNorm, LeakyReLU and the skip
// connection, which it reads from the staged rows (one read of x, one write).
//
// Numerics: B^T and A^T are signed sums (no constants); G carries two factors 1/2.  Against the fp64 result of the same fp32
// operands the outputs are closer than the direct kernels' (tests/test_gpu_kernels.py holds them to "not further than 2x").
#include "as_common.h"
#include "conv_epilogue.h"
#include "conv32_wino.h"
#include "conv32_wino_dev.h"
#ifndef WN_OLD_T
#define WN_OLD_T 0
#endif
#ifndef WN_OLD_Y
#define WN_OLD_Y 0
#endif

#define WN_SEG 64
#define WN_W 80                          // staged voxels per row: 8 + 64 + 8
#define WN_ROW_BYTES (WN_W * 128)        // 10,240
#define WN_COEF_OFF (4 * WN_ROW_BYTES)   // 40,960: scale, shift [2][32]
#define WN_X_OFF (WN_COEF_OFF + 768)     // (backward: k1, k2, k3, scale, shift, mean [6][32])  exchange [4 waves][2 j][4 g][64 lanes] float4
#define WN_LDS_BYTES (WN_X_OFF + 32768)  // 74,496
#ifndef WN_GRID
#define WN_GRID 512
#endif

struct WinoArgs {
  const float* zin;        // previous layer's pre-activation
  const float* ain;        // previous layer's input (skip connection) or null
  const float* in_scale;   // previous layer's BatchNorm as an affine
  const float* in_shift;
  float* a_out;            // by-product: the activated operand = previous layer's output
  const float* wq;         // transformed weights [16][4][64][4] (pack kind AS_PACK_WINO)
  EpilogueArgs ep;         // bias, z, moments
  PclDev g;
  int nseg, pairs;         // column segments per row; row pairs per (image, segment) over all combs
  float slope;
  // MODE 2 (data gradient): zin = z, ain = g_a, a_out = g_z (by-product), in_scale / in_shift = this layer's BatchNorm affine,
  // ep.z = g_x; plus
  const float* bn_mean;    // this layer's batch mean
  const float* bn_coef;    // stage-3 coefficients k1, k2, k3 [96]
  const float* nz;         // next BatchNorm backward (the layer below): pre-activation, affine, mean
  const float* n_scale;
  const float* n_shift;
  const float* n_mean;
  double* n_partial;       // [grid][64]: sum g_y, sum g_y * (z - mean) of the next BatchNorm
#ifdef WN_TIMING_BUILD
  long long* timing;       // diagnostic build only: [workgroup][wave][12] cycle counts per phase
#endif
};

// Diagnostic build (make EXTRA=-DWN_TIMING_BUILD): every wave adds up the shader cycles it spends in each phase; a launch
// with AS_WN_TIMING set dumps them to gpurun_out/wino_timing.bin (tests/tools/wino_microbench.py prints the averages).
#ifdef WN_TIMING_BUILD
#include <cstdio>
#include <cstdlib>
#define WN_T(slot) do { const long long now_ = clock64(); tacc[slot] += now_ - tlast; tlast = now_; } while (0)
#else
#define WN_T(slot) do { } while (0)
#endif

#define WN_FOR_8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

// MODE 0: training forward, no skip input; 1: training forward with skip input; 2: data gradient;
// 3: eval forward — the operand x is staged as is, out = lrelu((conv(x) + bias) * scale + shift) (+ x: the BasicBlock's skip
//    connection, read from the staged rows), nothing else written
// (diagnostic build -DWN_BWD_ONE_WG: the data gradient with one workgroup per CU, 512 registers per wave and every load requested
//  at the top of the tile — measured 397-435 us against 285-317 with two workgroups per CU: DESIGN 4)
#ifdef WN_BWD_ONE_WG
#define WN_WGS_PER_CU(MODE) ((MODE) == 2 ? 1 : 2)
#else
#define WN_WGS_PER_CU(MODE) 2
#endif
template <int MODE, int L>
__global__ __launch_bounds__(256, WN_WGS_PER_CU(MODE)) void conv32_wino_kernel(WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int d = 1 << L;
  constexpr bool SKIP = MODE == 1 || MODE == 2;             // a second tensor rides along with the operand rows
  constexpr bool BWD = MODE == 2;
  constexpr bool EVAL = MODE == 3;                          // operand staged as is; epilogue: affine + LeakyReLU (+ residual)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int H = p.g.H, W = p.g.W, Wp = p.g.Wp;

  // the wave's four transformed filters U[wave][c]: R[c][4q+e] = chunk q, element e
  f32x16 R[4];
  {
    const float* wb = p.wq + (wave * 4) * 1024 + lane * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(wb + c * 1024 + q * 256);
        R[c][4 * q + 0] = t4.x; R[c][4 * q + 1] = t4.y; R[c][4 * q + 2] = t4.z; R[c][4 * q + 3] = t4.w;
      }
  }
  const float bias_v = (!BWD && p.ep.bias) ? p.ep.bias[li] : 0.f;
  float bn_sc = 0.f, bn_sh = 0.f, bn_mu = 0.f, bn_dy = 0.f, bn_dx = 0.f;      // data gradient: next-BatchNorm sums per lane
  if constexpr (BWD) {
    float* tab = reinterpret_cast<float*>(smem + WN_COEF_OFF);
    const int i = threadIdx.x;
    if (i < 96) tab[i] = p.bn_coef[i];
    else if (i < 128) tab[i] = p.in_scale[i - 96];
    else if (i < 160) tab[i] = p.in_shift[i - 128];
    else if (i < 192) tab[i] = p.bn_mean[i - 160];
    bn_sc = p.n_scale[li]; bn_sh = p.n_shift[li]; bn_mu = p.n_mean[li];
  } else if (!EVAL && threadIdx.x < 64) {
    float* tab = reinterpret_cast<float*>(smem + WN_COEF_OFF);
    tab[threadIdx.x] = threadIdx.x < 32 ? p.in_scale[threadIdx.x] : p.in_shift[threadIdx.x - 32];
  }
  float ev_sc = 0.f, ev_sh = 0.f;
  if constexpr (EVAL) { ev_sc = p.ep.ep_scale[li]; ev_sh = p.ep.ep_shift[li]; }
  __syncthreads();

  // ---- row conversion: a row is staged d + 64 + d voxels wide (round 4: it was 8 + 64 + 8 whatever the dilation — 1.25 x the
  // operand tensors from HBM where a tile only reaches d voxels beyond its segment), a pair of rows per step, up to five 16-byte
  // chunks per thread (the fifth: 32 / 64 / 128 / 256 threads at d = 1 / 2 / 4 / 8); addressing: WnPair (conv32_wino_dev.h)
  using P = WnPair<256, L>;
  const int t = threadIdx.x;
  P pr;
  pr.init(t, 0);

  // ---- operand gather: this lane's tile li, input column m -> staged voxel 8 + c0 + (m-1) d; chunk 4h + q ----
  int op_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) op_off[m] = wn_addr<L>(8 + wn_c0<L>(li) + (m - 1) * d, 4 * h);
  // input rows (of the four of a tile) and sign of this wave's row transform: R = d[ra] + sg * d[rb]
  const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sg = wave == 1 ? 1.f : -1.f;
  // epilogue: this wave finishes output (oi, oj) of every tile
  const int oi = wave >> 1, oj = wave & 1;
  // sign of the second and third term of this wave's output row (wave-uniform: kept in a scalar register)
  const float sgy = __uint_as_float(__builtin_amdgcn_readfirstlane(oi == 0 ? 0x3f800000u : 0xbf800000u));
  const f32x4 sgy4 = {sgy, sgy, sgy, sgy};
  const unsigned io_off = (unsigned)((wn_c0<L>(4 * h) + oj * d) * 128 + 4 * li);
  const unsigned io_off2 = io_off + 4096u;                  // rows 8..15 of the accumulator layout: tiles +16 = 32 columns
  float st_c = 0.f, st_s1 = 0.f, st_s2 = 0.f, st_n = 0.f;      // BatchNorm moments of this lane's channel: shifted sums

#ifdef WN_TIMING_BUILD
  long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
  const long long wall0 = wall_clock64();
#endif
  const long t_total = (long)p.g.B * p.nseg * p.pairs;
  long t_next = t_total * blockIdx.x / gridDim.x;
  const long t_end = t_total * (blockIdx.x + 1) / gridDim.x;
  while (t_next < t_end) {
    const int blk = (int)(t_next / p.pairs);
    int pj0 = (int)(t_next - (long)blk * p.pairs);
    int r0 = 0, nrow = (H + d - 1) / d;                   // rows of comb r0
    while (pj0 >= (nrow + 1) / 2) { pj0 -= (nrow + 1) / 2; ++r0; nrow = (H - r0 + d - 1) / d; }
    const int pj1 = (int)min((long)((nrow + 1) / 2), pj0 + (t_end - t_next));
    t_next += pj1 - pj0;
    const int j0 = 2 * pj0, j1 = min(2 * pj1, nrow);
    const int seg = blk % p.nseg;
    const int b = blk / p.nseg;
    const int x_new = WN_SEG * seg;
    const int x0 = min(x_new, W - WN_SEG);
    // The last segment is shifted back to end at the image edge, so it shares columns with its neighbour — and a pixel's rounding
    // depends on the tile it falls into, which differs between the two (different tile origins): ONE of them must own a shared
    // column.  The neighbour keeps only its first `keep` columns, the shifted segment writes and counts all 64: the cut is an
    // UPPER bound, which the hardware applies for free — the output stores go through a buffer descriptor of keep * 128 bytes
    // per row and the out-of-range ones are dropped (a per-lane predicate on 16 stores cost 15 registers and spilled).
    const int keep = (seg == p.nseg - 2 && 64 * p.nseg > W) ? W - 64 * (p.nseg - 1) : WN_SEG;
    const long img = (long)b * p.g.Hp;
    const int px0 = x0 - 8 + p.g.pw;

    f32x4 pz[5], pa[5];
    static_assert(P::K == 5, "five chunks per thread and row pair");
    const bool edge = x0 < d || x0 + WN_SEG + d > W;        // (uniform) a halo column of this segment lies outside the image
    auto fetch_one = [&](int ja, const float* src, f32x4 (&pv)[5]) {     // rows ja, ja + 1; outside the image: any valid row
      const int ya = min(max(r0 + ja * d, 0), H - 1), yb = min(max(r0 + (ja + 1) * d, 0), H - 1);
      const float* base = src + ((img + ya + p.g.ph) * Wp + px0 + P::V0) * 32;     // row A's first staged voxel
      const long delta_f = (long)(yb - ya) * Wp * 32;                    // row A -> row B in floats (clamped rows: >= 0)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        // EVERY lane issues all five loads (the counted waits below rest on that): a lane without a fifth chunk re-reads the
        // pair's last one
        if (P::all_a(k)) {
          wn_load4(pv[k], base + 256 * 4 * k, pr.t16);
        } else if (P::all_b(k)) {
          const unsigned vo = P::full(k) ? pr.t16 : (pr.t <= P::CL ? pr.t16 : (unsigned)(P::CL * 16));
          wn_load4(pv[k], base + (256 * k - P::RC) * 4 + delta_f, vo);
        } else {                                           // the chunk that straddles the two rows: threads t >= TS are in row B
          wn_load4(pv[k], base + (256 * k - P::RC) * 4, pr.t16 + (pr.t >= P::TS ? (unsigned)(delta_f * 4) : (unsigned)(P::RC * 16)));
        }
      }
    };
    auto fetch_into = [&](int ja, f32x4 (&pz)[5], f32x4 (&pa)[5]) {
      fetch_one(ja, p.zin, pz);
      if (SKIP) fetch_one(ja, p.ain, pa);
    };
    auto convert_from = [&](int ja, f32x4 (&pz)[5], f32x4 (&pa)[5]) {    // -> ring slots (ja + 1) & 3, (ja + 2) & 3
      const float* tab = reinterpret_cast<const float*>(smem + WN_COEF_OFF) + (t & 7) * 4;
      f32x4 sc, sh, k1, k2, k3, bmu;
      if constexpr (BWD) {
        k1 = *reinterpret_cast<const f32x4*>(tab); k2 = *reinterpret_cast<const f32x4*>(tab + 32);
        k3 = *reinterpret_cast<const f32x4*>(tab + 64); sc = *reinterpret_cast<const f32x4*>(tab + 96);
        sh = *reinterpret_cast<const f32x4*>(tab + 128); bmu = *reinterpret_cast<const f32x4*>(tab + 160);
      } else if constexpr (!EVAL) {
        sc = *reinterpret_cast<const f32x4*>(tab); sh = *reinterpret_cast<const f32x4*>(tab + 32);
      }
      // per row of the pair (uniform): inside the image?  one of this piece's own rows (by-product)?  ring slot, output row
      const int y_a = r0 + ja * d, y_b = r0 + (ja + 1) * d;
      const bool in_a = y_a >= 0 && y_a < H, in_b = y_b >= 0 && y_b < H;
      const bool own_a = ja >= j0 && ja < j1, own_b = ja + 1 >= j0 && ja + 1 < j1;
      const int ya_c = min(max(y_a, 0), H - 1), yb_c = min(max(y_b, 0), H - 1);
      float* out_a = p.a_out + ((img + ya_c + p.g.ph) * Wp + px0 + P::V0) * 32;     // (an own row is never a clamped one)
      const long delta_f = (long)(yb_c - ya_c) * Wp * 32;
      char* ring_a = smem + ((ja + 1) & 3) * WN_ROW_BYTES;
      char* ring_b = smem + ((ja + 2) & 3) * WN_ROW_BYTES;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const bool strad = !P::all_a(k) && !P::all_b(k);
        const bool rowb = pr.in_b(k);                       // (per lane only for the straddling chunk)
        f32x4 yv;
        if constexpr (BWD) {                                // stage 3 of the BatchNorm backward (conv32_bwd.hip's arithmetic)
          const f32x4 ga = pa[k], zz = pz[k];
          const f32x4 yy = zz * sc + sh;
          const f32x4 gl = ga * p.slope;
          f32x4 gy;
          gy.x = yy.x > 0.f ? ga.x : gl.x; gy.y = yy.y > 0.f ? ga.y : gl.y;
          gy.z = yy.z > 0.f ? ga.z : gl.z; gy.w = yy.w > 0.f ? ga.w : gl.w;
          yv = (gy - k1 - (zz - bmu) * k2) * k3;
        } else if constexpr (EVAL) {
          yv = pz[k];                                       // (the tensor's halo is zero in memory: no column mask either)
        } else {
          yv = pz[k] * sc + sh;
          const f32x4 ys = yv * p.slope;                    // 0 < slope < 1: lrelu(y) = max(y, slope*y)
          yv.x = fmaxf(yv.x, ys.x); yv.y = fmaxf(yv.y, ys.y);
          yv.z = fmaxf(yv.z, ys.z); yv.w = fmaxf(yv.w, ys.w);
          if (SKIP) yv += pa[k];
        }
        // halo voxels: only they can lie outside the image (then they read as zero padding), and they are not part of the
        // by-product; which chunks can hold any is known at compile time
        bool halo = false;
        if (P::may_halo(k, false) || P::may_halo(k, true)) {
          const int vq = pr.vq(k, rowb);
          halo = vq < d || vq >= WN_SEG + d;
          if (!EVAL && edge && halo) {
            const int xx = x0 - d + vq;
            if (!(xx >= 0 && xx < W)) yv = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
        if (!(rowb ? in_b : in_a)) yv = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool act = pr.active(k);
        if (act) *reinterpret_cast<f32x4*>((rowb ? ring_b : ring_a) + (strad ? (rowb ? pr.lds_b(k) : pr.lds_a(k)) : (P::all_b(k) ? pr.lds_b(k) : pr.lds_a(k)))) = yv;
        if (!EVAL && act && (rowb ? own_b : own_a) && !halo) {
          if (P::all_a(k)) wn_store4(out_a + 256 * 4 * k, pr.t16, yv);
          else if (P::all_b(k)) wn_store4(out_a + (256 * k - P::RC) * 4 + delta_f, pr.t16, yv);
          else wn_store4(out_a + (256 * k - P::RC) * 4, pr.t16 + (rowb ? (unsigned)(delta_f * 4) : (unsigned)(P::RC * 16)), yv);
        }
      }
    };
    auto wait_all = [&](f32x4 (&pz)[5], f32x4 (&pa)[5]) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]) :: "memory");
      if (SKIP) asm volatile("" : "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]), "+v"(pa[4]) :: "memory");
    };

    WN_T(0);
    // ---- run-in: activated rows j0-1 .. j0+2 (two pairs in flight; the loads of the second pair retire after the first's) ----
    {
      f32x4 qz[5], qa[5];
      fetch_into(j0 - 1, pz, pa); fetch_into(j0 + 1, qz, qa);
      if (SKIP) asm volatile("s_waitcnt vmcnt(10)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]) :: "memory");
      if (SKIP) asm volatile("" : "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]), "+v"(pa[4]) :: "memory");
      convert_from(j0 - 1, pz, pa);
      wait_all(qz, qa);
      convert_from(j0 + 1, qz, qa);
    }
    __syncthreads();
    WN_T(1);

    for (int j = j0; j < j1; j += 2) {
      const bool more = j + 2 < j1;                        // (the last tile of a piece converts nothing: its fetch stays — a
                                                           //  conditional fetch made hipcc shuffle the load registers)
      fetch_one(j + 3, p.zin, pz);                         // in flight during the matrix phase (the skip rows follow it:
                                                           // ten more registers across the matrix phase spilled)
#define WN_IMM(r) (wn_c0<L>(((r) & 3) + 8 * (((r) >> 2) & 1)) * 128)
#define WN_COL(r) (wn_c0<L>(((r) & 3) + 8 * ((r) >> 2)) + wn_c0<L>(4 * h) + oj * d)    /* column of accumulator row r in the segment */
      const int yrow = j + oi;
      const bool row_ok = yrow < j1;                       // (wave-uniform) the pair's second row may lie outside the piece
      float res[16], zt[16];                                // data gradient: g_a (skip connection) and the next layer's
                                                            // pre-activation at this wave's output pixels
      auto late_loads = [&]() {
        if (SKIP) fetch_one(j + 3, p.ain, pa);
        if constexpr (BWD) {
          if (row_ok) {
            const long ovox = ((img + r0 + yrow * d + p.g.ph) * Wp + x0 + p.g.pw) * 32;
#define WN_LD(r) wn_load_imm<WN_IMM(r)>(res[r], p.ain + ovox, io_off); wn_load_imm<WN_IMM(r)>(res[8 + r], p.ain + ovox, io_off2); \
                 wn_load_imm<WN_IMM(r)>(zt[r], p.nz + ovox, io_off); wn_load_imm<WN_IMM(r)>(zt[8 + r], p.nz + ovox, io_off2);
            WN_FOR_8(WN_LD)
#undef WN_LD
          }
        }
      };
      if constexpr (WN_WGS_PER_CU(MODE) == 1) late_loads();  // (one workgroup per CU: 512 registers, everything requested up front)
      const char* row_a = smem + ((j + ra) & 3) * WN_ROW_BYTES;   // input row m of the tile = comb row j-1+m = slot (j+m) & 3
      const char* row_b = smem + ((j + rb) & 3) * WN_ROW_BYTES;
      f32x16 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      f32x4 xa[4], xb[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        xa[m] = *reinterpret_cast<const f32x4*>(row_a + op_off[m]);
        xb[m] = *reinterpret_cast<const f32x4*>(row_b + op_off[m]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 V[4];
        {
          f32x4 Rt[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) Rt[m] = xa[m] + sg * xb[m];
          // (packed differences: -4 % on the forward, -1.5 % on the inference block; the data gradient's instantiation, at the
          //  256-register limit already, pays for the register pairs with 8-10 spilled registers: +2 %, so not there)
          V[0] = wn_sub4_if<!BWD>(Rt[0], Rt[2]); V[1] = Rt[1] + Rt[2]; V[2] = wn_sub4_if<!BWD>(Rt[2], Rt[1]);
          V[3] = wn_sub4_if<!BWD>(Rt[1], Rt[3]);
          if constexpr (!BWD) wn_before_mfma();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < 4) {                                   // the next chunk's operands: in flight under this chunk's MFMAs
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            xa[m] = *reinterpret_cast<const f32x4*>(row_a + (op_off[m] ^ ((q + 1) << 4)));
            xb[m] = *reinterpret_cast<const f32x4*>(row_b + (op_off[m] ^ ((q + 1) << 4)));
          }
        }
#pragma unroll
#ifdef WN_EXP_NOMFMA
        for (int c = 0; c < 4; ++c) acc[c][0] += V[c].x * R[c][4 * q + 0];   // (diagnostic build: no matrix instructions; results are wrong)
#else
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[c].x, R[c][4 * q + 0], acc[c], 0, 0, 0);
#endif
#pragma unroll
#ifdef WN_EXP_NOMFMA
        for (int c = 0; c < 4; ++c) acc[c][0] += V[c].y * R[c][4 * q + 1];   // (diagnostic build: no matrix instructions; results are wrong)
#else
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[c].y, R[c][4 * q + 1], acc[c], 0, 0, 0);
#endif
#pragma unroll
#ifdef WN_EXP_NOMFMA
        for (int c = 0; c < 4; ++c) acc[c][0] += V[c].z * R[c][4 * q + 2];   // (diagnostic build: no matrix instructions; results are wrong)
#else
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[c].z, R[c][4 * q + 2], acc[c], 0, 0, 0);
#endif
#pragma unroll
#ifdef WN_EXP_NOMFMA
        for (int c = 0; c < 4; ++c) acc[c][0] += V[c].w * R[c][4 * q + 3];   // (diagnostic build: no matrix instructions; results are wrong)
#else
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[c].w, R[c][4 * q + 3], acc[c], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      WN_T(2);
      // T[j] = (M[r] A)[j]: A^T = [1 1 1 0; 0 1 -1 -1]
      {
        char* xw = smem + WN_X_OFF + (wave * 2) * 4096 + lane * 16;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          f32x4 t0, t1;
          if constexpr (BWD || WN_OLD_T) {                 // (element by element, as it was: this instantiation has no register to spare)
            t0.x = (acc[0][4 * gq + 0] + acc[1][4 * gq + 0]) + acc[2][4 * gq + 0];
            t0.y = (acc[0][4 * gq + 1] + acc[1][4 * gq + 1]) + acc[2][4 * gq + 1];
            t0.z = (acc[0][4 * gq + 2] + acc[1][4 * gq + 2]) + acc[2][4 * gq + 2];
            t0.w = (acc[0][4 * gq + 3] + acc[1][4 * gq + 3]) + acc[2][4 * gq + 3];
            t1.x = (acc[1][4 * gq + 0] - acc[2][4 * gq + 0]) - acc[3][4 * gq + 0];
            t1.y = (acc[1][4 * gq + 1] - acc[2][4 * gq + 1]) - acc[3][4 * gq + 1];
            t1.z = (acc[1][4 * gq + 2] - acc[2][4 * gq + 2]) - acc[3][4 * gq + 2];
            t1.w = (acc[1][4 * gq + 3] - acc[2][4 * gq + 3]) - acc[3][4 * gq + 3];
          } else {
            if (gq == 0) wn_after_mfma();                  // (the matrix phase ended at a sched_barrier right above)
            f32x4 a4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) a4[c] = (f32x4){acc[c][4 * gq + 0], acc[c][4 * gq + 1], acc[c][4 * gq + 2], acc[c][4 * gq + 3]};
            t0 = (a4[0] + a4[1]) + a4[2];
            t1 = wn_sub4(wn_sub4(a4[1], a4[2]), a4[3]);
          }
          *reinterpret_cast<f32x4*>(xw + gq * 1024) = t0;
          *reinterpret_cast<f32x4*>(xw + 4096 + gq * 1024) = t1;
        }
      }
      if constexpr (WN_WGS_PER_CU(MODE) != 1) late_loads();  // (the accumulators are dead: registers to spare)
      if constexpr (EVAL) {
        // the skip connection of the block = the operand at the output pixels: in the staged rows j, j+1 — read before B1
        // (afterwards other waves overwrite those slots)
        if (p.ep.residual != nullptr) {
          const char* rrow = smem + ((j + oi + 1) & 3) * WN_ROW_BYTES + (li & 3) * 4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tl = (r & 3) + 8 * (r >> 2);          // + 4h: this lane's tile of accumulator row r
            res[r] = *reinterpret_cast<const float*>(rrow + wn_addr<L>(8 + wn_c0<L>(tl) + wn_c0<L>(4 * h) + oj * d, li >> 2));
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) res[r] = 0.f;
        }
      }
      int keep_l = keep;                                   // (opaque: hipcc otherwise hoists the 16 shared-column masks of the
      asm volatile("" : "+s"(keep_l));                     //  moments out of the tile loop, into 16 registers)
      WN_T(3);
      __syncthreads();                                     // B1: the T tiles are in place; nobody reads rows j-1, j any more
      WN_T(4);
      // ---- Y[oi][oj] = (A^T T)[oi][oj]: waves (oi, oi+1, oi+2) with signs (+,+,+) / (+,-,-) ----
      f32x16 Y;
      {
        const char* xr = smem + WN_X_OFF + (oi * 2 + oj) * 4096 + lane * 16;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 u0 = *reinterpret_cast<const f32x4*>(xr + gq * 1024);
          const f32x4 u1 = *reinterpret_cast<const f32x4*>(xr + 8192 + gq * 1024);
          const f32x4 u2 = *reinterpret_cast<const f32x4*>(xr + 16384 + gq * 1024);
          // (u0 + u1) + u2 for the first output row, (u0 - u1) - u2 for the second: one fma per term with s = +-1 (the same
          // roundings as the add / subtract, no select between two results)
          f32x4 yv;
          if constexpr (BWD || WN_OLD_Y) yv = oi == 0 ? (u0 + u1) + u2 : (u0 - u1) - u2;
          else yv = __builtin_elementwise_fma(u2, sgy4, __builtin_elementwise_fma(u1, sgy4, u0));
          Y[4 * gq + 0] = yv.x + bias_v; Y[4 * gq + 1] = yv.y + bias_v; Y[4 * gq + 2] = yv.z + bias_v; Y[4 * gq + 3] = yv.w + bias_v;
        }
      }
      WN_T(5);
      if constexpr (BWD) {
        // every load so far is home (operand rows, g_a / next-z at the output pixels); the stores below are younger
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7]),
                       "+v"(res[8]), "+v"(res[9]), "+v"(res[10]), "+v"(res[11]), "+v"(res[12]), "+v"(res[13]), "+v"(res[14]),
                       "+v"(res[15]) :: "memory");
        asm volatile("" : "+v"(zt[0]), "+v"(zt[1]), "+v"(zt[2]), "+v"(zt[3]), "+v"(zt[4]), "+v"(zt[5]), "+v"(zt[6]), "+v"(zt[7]),
                          "+v"(zt[8]), "+v"(zt[9]), "+v"(zt[10]), "+v"(zt[11]), "+v"(zt[12]), "+v"(zt[13]), "+v"(zt[14]),
                          "+v"(zt[15]) :: "memory");
        asm volatile("" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]), "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]),
                          "+v"(pa[3]), "+v"(pa[4]) :: "memory");
        if (row_ok) {
          const int y = r0 + yrow * d;
          float* gx_base = p.ep.z + ((img + y + p.g.ph) * Wp + x0 + p.g.pw) * 32;
#pragma unroll
          for (int r = 0; r < 16; ++r) Y[r] += res[r];
          {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(gx_base, 0, keep_l * 128, 0x00020000);
#define WN_ST(r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Y[r]), rs, (int)io_off + WN_IMM(r), 0, 0); \
                 __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Y[8 + r]), rs, (int)io_off2 + WN_IMM(r), 0, 0);
            WN_FOR_8(WN_ST)
#undef WN_ST
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float yv = fmaf(zt[r], bn_sc, bn_sh);
            float gy = yv > 0.f ? Y[r] : Y[r] * p.slope;
            if (keep_l < WN_SEG) gy = WN_COL(r) < keep_l ? gy : 0.f;
            bn_dy += gy; bn_dx = fmaf(gy, zt[r] - bn_mu, bn_dx);
          }
        }
      }
      if constexpr (!BWD) {
        // The operand rows of the next tile are home BEFORE this tile's stores go out: one unconditional wait on the very
        // registers the loads were issued into.  (First version: a counted wait behind the 16 stores, vmcnt(16) or vmcnt(0) by
        // branch — hipcc merged the two asm statements' register operands by COPYING the load destinations in front of the
        // branch, i.e. before the wait: rare stale operands, different from process to process.  tests/tools/check_async_loads.py
        // scans the ISA for any read of an in-flight load destination.)
        wait_all(pz, pa);
      }
      if (!BWD && row_ok) {
        const int y = r0 + yrow * d;
        float* z_base = p.ep.z + ((img + y + p.g.ph) * Wp + x0 + p.g.pw) * 32;
        if constexpr (EVAL) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {                    // conv_epilogue's arithmetic (epilogue 1)
            float yv = Y[r] * ev_sc + ev_sh;
            yv = fmaxf(yv, yv * p.slope);                   // 0 < slope < 1 (checked by the entry point): lrelu(y), the same bits,
            Y[r] = yv + res[r];                             // two instructions instead of three
          }
        }
        {
          const auto rs = __builtin_amdgcn_make_buffer_rsrc(z_base, 0, keep_l * 128, 0x00020000);
#define WN_ST(r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Y[r]), rs, (int)io_off + WN_IMM(r), 0, 0); \
                 __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Y[8 + r]), rs, (int)io_off2 + WN_IMM(r), 0, 0);
          WN_FOR_8(WN_ST)
#undef WN_ST
        }
        if (!EVAL && p.ep.stat_mean != nullptr) {
          if (keep_l >= WN_SEG) {
            st_c = st_n == 0.f ? Y[0] : st_c;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float dd = Y[r] - st_c; st_s1 += dd; st_s2 = fmaf(dd, dd, st_s2); }
            st_n += 16.f;
          } else {
            st_c = st_n == 0.f ? Y[0] : st_c;              // (any value will do as the pivot)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool in = WN_COL(r) < keep_l;
              const float dd = in ? Y[r] - st_c : 0.f;
              st_s1 += dd; st_s2 = fmaf(dd, dd, st_s2); st_n += in ? 1.f : 0.f;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      WN_T(6);
      WN_T(7);
      if (more) convert_from(j + 3, pz, pa);
      WN_T(8);
      __syncthreads();                                     // B2: activated rows j+3, j+4 are in place; the exchange is free
      WN_T(9);
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef WN_TIMING_BUILD
  WN_T(10);
  if (p.timing && lane == 0) {
    long long* o = p.timing + ((long)blockIdx.x * 4 + wave) * 12;
    for (int i = 0; i < 11; ++i) o[i] = tacc[i];
    o[11] = wall_clock64() - wall0;
  }
#endif
  if constexpr (BWD) {
    // next-BatchNorm sums: 8 (wave, half) partials per channel -> one fp64 pair per workgroup
    float* scr = reinterpret_cast<float*>(smem);        // [8][2][32]
    scr[((wave * 2 + h) * 2 + 0) * 32 + li] = bn_dy;
    scr[((wave * 2 + h) * 2 + 1) * 32 + li] = bn_dx;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, cch = threadIdx.x & 31;
      double sum = 0.0;
      for (int q = 0; q < 8; ++q) sum += (double)scr[(q * 2 + which) * 32 + cch];
      p.n_partial[(long)blockIdx.x * 64 + which * 32 + cch] = sum;
    }
    return;
  }
  if (p.ep.stat_mean != nullptr) {
    // lane sums -> (n, mean, M2) -> one partial per workgroup: 8 (wave, half) partials per channel, merged in fixed order
    float* part = reinterpret_cast<float*>(smem);      // [8][32][3]
    const float mean_l = st_n > 0.f ? st_c + st_s1 / st_n : 0.f;
    const float m2_l = st_n > 0.f ? fmaxf(st_s2 - st_s1 * st_s1 / st_n, 0.f) : 0.f;
    float* mine = part + ((wave * 2 + h) * 32 + li) * 3;
    mine[0] = st_n; mine[1] = mean_l; mine[2] = m2_l;
    __syncthreads();
    if (threadIdx.x < 32) {
      TileStats run; run.n = 0.f; run.mean = 0.f; run.m2 = 0.f;
      for (int q = 0; q < 8; ++q) {
        TileStats ts;
        ts.n = part[(q * 32 + li) * 3]; ts.mean = part[(q * 32 + li) * 3 + 1]; ts.m2 = part[(q * 32 + li) * 3 + 2];
        stats_merge(run, ts);
      }
      stats_write(p.ep, blockIdx.x, run);
    }
  }
}

static int wn_grid(int mode) { return WN_GRID / (2 / WN_WGS_PER_CU(mode)); }
static int wn_log2(int d) { return d == 1 ? 0 : d == 2 ? 1 : d == 4 ? 2 : d == 8 ? 3 : -1; }

static long wn_pairs(int H, int d) {
  long n = 0;
  for (int r = 0; r < d; ++r) n += ((H - r + d - 1) / d + 1) / 2;
  return n;
}

// Applicable to the refinement geometry: 2-D 3x3 stride 1, dilation 1/2/4/8 within the halo, rows of at least 64 pixels and
// enough row pairs for the fixed grid.
bool conv32_wino_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (s->kd != 1 || s->kh != 3 || s->kw != 3 || s->stride != 1) return false;
  if (wn_log2(s->dil) < 0 || s->pad_h != s->dil || s->pad_w != s->dil) return false;
  if (gin->D != 1 || gout->D != 1 || gin->pd != 0) return false;
  if (gin->B != gout->B || gin->H != gout->H || gin->W != gout->W) return false;
  if (gin->ph != gout->ph || gin->pw != gout->pw || gin->pw < 8 || gin->ph < s->dil) return false;
  if (gout->W < WN_SEG || gout->H < 2 * s->dil) return false;
  const long units = (long)gout->B * ((gout->W + WN_SEG - 1) / WN_SEG) * wn_pairs(gout->H, s->dil);
  return units >= 4L * WN_GRID && units < (1L << 31);
}

int conv32_wino_parts(void) { return WN_GRID; }
int conv32_wino_dgrad_parts(void) { return wn_grid(2); }

template <int MODE> static const void* wn_kernel(int L) {
  switch (L) {
    case 0: return reinterpret_cast<const void*>(conv32_wino_kernel<MODE, 0>);
    case 1: return reinterpret_cast<const void*>(conv32_wino_kernel<MODE, 1>);
    case 2: return reinterpret_cast<const void*>(conv32_wino_kernel<MODE, 2>);
    default: return reinterpret_cast<const void*>(conv32_wino_kernel<MODE, 3>);
  }
}

static int wn_launch(int mode, int L, const WinoArgs& a, const char* who, void* stream) {
  static AsPerDevice attr_set[16];
  const void* fn = mode == 3 ? wn_kernel<3>(L) : (mode == 2 ? wn_kernel<2>(L) : (mode == 1 ? wn_kernel<1>(L) : wn_kernel<0>(L)));
  const int fi = mode * 4 + L;
  if (!attr_set[fi].get()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_BYTES);
    if (e != hipSuccess) { as_set_error("%s: %s", who, hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set[fi].set();
  }
  WinoArgs args = a;
#ifdef WN_TIMING_BUILD
  static long long* timing_buf = nullptr;
  const size_t timing_bytes = (size_t)WN_GRID * 4 * 12 * 8;
  if (!timing_buf) hipMalloc(&timing_buf, timing_bytes);
  hipMemsetAsync(timing_buf, 0, timing_bytes, (hipStream_t)stream);
  args.timing = timing_buf;
#endif
  void* kargs[] = {&args};
  hipError_t le = hipLaunchKernel(fn, dim3(wn_grid(mode)), dim3(256), kargs, WN_LDS_BYTES, (hipStream_t)stream);
  if (le != hipSuccess) { as_set_error("%s: launch failed: %s", who, hipGetErrorString(le)); return AS_ERR_LAUNCH; }
#ifdef WN_TIMING_BUILD
  if (getenv("AS_WN_TIMING")) {                            // dump THIS launch (synchronises: diagnostic build only)
    hipStreamSynchronize((hipStream_t)stream);
    void* hbuf = malloc(timing_bytes); hipMemcpy(hbuf, timing_buf, timing_bytes, hipMemcpyDeviceToHost);
    char name[128]; snprintf(name, sizeof name, "gpurun_out/wino_timing_m%d.bin", mode);
    FILE* f = fopen(name, "wb"); if (f) { fwrite(hbuf, 1, timing_bytes, f); fclose(f); } free(hbuf);
  }
#endif
  return AS_OK;
}

int conv32_wino_launch(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift, float* a_out,
                       const as_pcl* g, const as_conv_shape* s, const float* wino_w, const float* bias, float slope,
                       float* z, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  WinoArgs a = {};
  a.zin = z_prev; a.ain = a_prevprev; a.in_scale = in_scale; a.in_shift = in_shift; a.a_out = a_out; a.wq = wino_w;
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = nullptr; a.ep.ep_shift = nullptr; a.ep.residual = nullptr;
  a.ep.stat_mean = stat_mean; a.ep.stat_m2 = stat_m2; a.ep.stat_cnt = stat_cnt; a.ep.epilogue = 0; a.ep.slope = slope;
  a.g = as_make_dev(g);
  a.nseg = (g->W + WN_SEG - 1) / WN_SEG; a.pairs = (int)wn_pairs(g->H, s->dil); a.slope = slope;
  return wn_launch(a_prevprev != nullptr ? 1 : 0, wn_log2(s->dil), a, "as_conv32_wino_fwd", stream);
}

// Eval forward (MODE 3): out = lrelu((conv(x) + bias) * scale + shift) (+ x if residual).
int conv32_wino_eval_launch(const float* x, const as_pcl* g, const as_conv_shape* s, const float* wino_w, const float* bias,
                            const float* scale, const float* shift, float slope, int residual, float* out, void* stream) {
  WinoArgs a = {};
  a.zin = x; a.wq = wino_w;
  a.ep.bias = bias; a.ep.z = out; a.ep.ep_scale = scale; a.ep.ep_shift = shift; a.ep.residual = residual ? x : nullptr;
  a.ep.epilogue = 1; a.ep.slope = slope;
  a.g = as_make_dev(g);
  a.nseg = (g->W + WN_SEG - 1) / WN_SEG; a.pairs = (int)wn_pairs(g->H, s->dil); a.slope = slope;
  return wn_launch(3, wn_log2(s->dil), a, "as_conv32_wino_eval", stream);
}

// Data gradient of the layer (MODE 2): g_z (by-product), g_x = dgrad(g_z) + g_a, next-BatchNorm sums [WN_GRID][64].
int conv32_wino_dgrad_launch(const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s, const float* wino_wt,
                             const float* scale, const float* shift, const float* mean, const float* coef, float slope,
                             const float* next_z, const float* next_scale, const float* next_shift, const float* next_mean,
                             float* g_z, float* g_x, double* next_partial, void* stream) {
  WinoArgs a = {};
  a.zin = z; a.ain = g_a; a.in_scale = scale; a.in_shift = shift; a.a_out = g_z; a.wq = wino_wt;
  a.ep.z = g_x; a.ep.slope = slope;
  a.g = as_make_dev(g);
  a.nseg = (g->W + WN_SEG - 1) / WN_SEG; a.pairs = (int)wn_pairs(g->H, s->dil); a.slope = slope;
  a.bn_mean = mean; a.bn_coef = coef; a.nz = next_z; a.n_scale = next_scale; a.n_shift = next_shift; a.n_mean = next_mean;
  a.n_partial = next_partial;
  return wn_launch(2, wn_log2(s->dil), a, "as_conv32_wino_bwd", stream);
}
