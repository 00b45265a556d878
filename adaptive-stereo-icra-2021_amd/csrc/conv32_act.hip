// Training forward of a full-resolution refinement layer (3x3, any dilation <= 8, stride 1, 32->32: stereo_net.py:10-18,
// 33-51, 97) that APPLIES THE PREVIOUS LAYER'S BatchNorm + LeakyReLU (+ skip connection) to its operand on the way in:
//   a_prev = lrelu(z_prev * scale + shift) (+ a_prevprev)        formed in LDS, written back once as a by-product
//   z      = conv(a_prev) + bias, BatchNorm moments of z           (what conv32_lds_kernel<0,false> does)
// Before: bn_act_fwd_kernel (read z_prev, read a_prevprev, write a_prev: 95 us at 4 pairs, HBM-bound) and then
// conv32_lds_kernel (read a_prev, write z: 290 us, matrix-bound).  Here the matrix-bound kernel carries the element-wise
// pass: 4 tensor passes instead of 5 and one launch instead of two.
//
// Structure (the fused backward's, conv32_bwd.hip): a workgroup walks a COMB of rows y = r, r+d, r+2d, .. of a 128-pixel
// column segment, so the three input rows of a tile (y-d, y, y+d) are the previous, current and next row of a ring of
// three activated rows in LDS, and every z_prev / a_prevprev row is fetched, activated and stored exactly once per comb
// piece (+2 rows of run-in).  Per tile, two barriers:
//   top      all four waves request the z_prev / a_prevprev chunks of row j+2 (plain coalesced loads into registers: the
//            latency hides behind the matrix phase)
//   matrix   144 MFMAs per wave (32 pixels x 32 channels; five taps' B fragments resident in 80 registers, four streamed from
//            LDS one chunk ahead: nine resident plus the 40-register prefetch did not fit in 256)
//   B1       everybody is done with row j-1's slot
//   vector   wait for the prefetch; the 16 output stores and the BatchNorm moments of the tile; row j+2: activate, write the
//            by-product, store swizzled into the freed slot
//   B2
// Every vector-memory instruction of the loop is inline asm with explicit waits: vmcnt retires in order, and a compiler-
// counted wait for a load that is older than stores the compiler cannot see would wait for those stores too.
// Measured at 4 pairs: 328-336 us per layer (65-67 % of the fp32 matrix peak) against 411-425 us for the two launches.
#include "as_common.h"
#include "conv_epilogue.h"
#include "conv32_act.h"

#define CA_W 144                        // staged voxels per row: 8 + 128 + 8
#define CA_ROW_BYTES (CA_W * 128)       // 18,432
#define CA_COEF_OFF (3 * CA_ROW_BYTES)  // 55,296: scale, shift [2][32]
#define CA_WLDS_OFF (CA_COEF_OFF + 256) // 55,552: B fragments of the taps that are not register-resident [2][4][64][4]
#define CA_RES 5                        // taps resident in registers (80); with 7 and the 40-register prefetch hipcc spilled 48 registers
#define CA_LDS_BYTES (CA_WLDS_OFF + (9 - CA_RES) * 4096)   // 71,936
#ifndef CA_GRID
#define CA_GRID 512                     // two resident workgroups per CU (tests/tools/grid_sweep.sh rebuilds with EXTRA=-DCA_GRID=n)
#endif
static int ca_grid(void) { return CA_GRID; }

struct ActArgs {
  const float* zin;        // previous layer's pre-activation
  const float* ain;        // previous layer's input (skip connection) or null
  const float* in_scale;   // previous layer's BatchNorm as an affine
  const float* in_shift;
  float* a_out;            // by-product: the activated operand = previous layer's output
  const float* wq;         // packed weights [9][4][64][4] (as_conv32_pack_weights, transpose_flip = 0)
  EpilogueArgs ep;         // bias, z, moments
  PclDev g;
  int dil, nseg;
  float slope;
};

__device__ inline void ca_load4(f32x4& v, const float* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
}
// (a store of more than 64 bits reads its data registers late: the instruction that overwrites them must be at least two
// wait states away — hipcc inserts that for its own stores, not behind inline asm)
__device__ inline void ca_store4(float* sbase, unsigned voff, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <int IMM> __device__ inline void ca_store_imm(float* sbase, unsigned voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
#define CA_ROW_IMM(r) ((((r) & 3) + 8 * ((r) >> 2)) * 128)
#define CA_FOR_ROWS(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

__device__ inline f32x4 ca_chunk(const char* row, int v, int h, int q) {
  return *reinterpret_cast<const f32x4*>(row + v * 128 + (((4 * h + q) ^ ((v >> 1) & 7)) << 4));
}

template <bool SKIP>
__global__ __launch_bounds__(256, 2) void conv32_act_kernel(ActArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int H = p.g.H, W = p.g.W, Wp = p.g.Wp, d = p.dil;

  // B fragments of the first CA_RES taps: R[tap][4q+e] = chunk q, element e; the other taps' live in LDS
  f32x16 R[CA_RES];
  {
    const float* wb = p.wq + lane * 4;
#pragma unroll
    for (int tp = 0; tp < CA_RES; ++tp)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(wb + tp * 1024 + q * 256);
        R[tp][4 * q + 0] = t4.x; R[tp][4 * q + 1] = t4.y; R[tp][4 * q + 2] = t4.z; R[tp][4 * q + 3] = t4.w;
      }
  }
  const float bias_v = p.ep.bias ? p.ep.bias[li] : 0.f;
#pragma unroll
  for (int i = 0; i < 9 - CA_RES; ++i)                     // 1,024 floats per tap, one float4 per thread
    *reinterpret_cast<f32x4*>(smem + CA_WLDS_OFF + i * 4096 + threadIdx.x * 16) =
        *reinterpret_cast<const f32x4*>(p.wq + (CA_RES + i) * 1024 + threadIdx.x * 4);
  if (threadIdx.x < 64) {
    float* tab = reinterpret_cast<float*>(smem + CA_COEF_OFF);
    tab[threadIdx.x] = threadIdx.x < 32 ? p.in_scale[threadIdx.x] : p.in_shift[threadIdx.x - 32];
  }
  __syncthreads();
  // row conversion: thread t owns chunks f = t + 256k, k < 5 (k = 4: t < 128) of a 144-voxel row: voxel (t >> 3) + 32k,
  // channel group t & 7; the swizzle term (v >> 1) & 7 = (t >> 4) & 7 is the same for every k
  const int t = threadIdx.x;
  const int c4 = (t & 7) * 4;
  const int cv_swz = (t >> 3) * 128 + (((t & 7) ^ ((t >> 4) & 7)) << 4);
  const unsigned io_off = (unsigned)(512 * h + 4 * li);
  float st_c = 0.f, st_s1 = 0.f, st_s2 = 0.f, st_n = 0.f;      // BatchNorm moments of this lane's channel: shifted sums

  // work split: see conv32_bwd.hip (equal ranges of the launch's tiles, one to three comb pieces per workgroup)
  const long t_total = (long)p.g.B * p.nseg * H;
  long t_next = t_total * blockIdx.x / gridDim.x;
  const long t_end = t_total * (blockIdx.x + 1) / gridDim.x;
  while (t_next < t_end) {
    const int blk = (int)(t_next / H);
    int j0 = (int)(t_next - (long)blk * H);
    int r0 = 0, nrow = (H + d - 1) / d;                   // rows of comb r0
    while (j0 >= nrow) { j0 -= nrow; ++r0; nrow = (H - r0 + d - 1) / d; }
    const int j1 = (int)min((long)nrow, j0 + (t_end - t_next));
    t_next += j1 - j0;
    const int seg = blk % p.nseg;
    const int b = blk / p.nseg;
    const int x_new = 128 * seg;
    const int x0 = min(x_new, W - 128);
    const long img = (long)b * p.g.Hp;
    const int px0 = x0 - 8 + p.g.pw;

    // every vector-memory instruction of the tile loop is inline asm with explicit waits: hipcc's own wait bookkeeping
    // cannot see asm stores and would turn a wait for an older load into a wait for them (vmcnt retires in order)
    f32x4 pz[5], pa[5];
    const unsigned t16 = (unsigned)t * 16u;
    auto fetch_into = [&](int jj, f32x4 (&pz)[5], f32x4 (&pa)[5]) {     // rows outside the image: any valid row (-> zeros)
      const int y = min(max(r0 + jj * d, 0), H - 1);
      const long off = ((img + y + p.g.ph) * Wp + px0) * 32;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        ca_load4(pz[k], p.zin + off + k * 1024, t16);
        if (SKIP) ca_load4(pa[k], p.ain + off + k * 1024, t16);
      }
    };
    auto fetch_row = [&](int jj) { fetch_into(jj, pz, pa); };
    auto wait_row = [&]() {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]) :: "memory");
      if (SKIP) asm volatile("" : "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]), "+v"(pa[4]) :: "memory");
    };
    auto convert_from = [&](int jj, f32x4 (&pz)[5], f32x4 (&pa)[5]) {   // -> ring slot (jj + 1) % 3, swizzled; by-product for own rows
      const int y = r0 + jj * d;
      char* dst = smem + ((jj + 1) % 3) * CA_ROW_BYTES + cv_swz;
      if (y < 0 || y >= H) {                               // (workgroup-uniform)
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dst + k * 4096) = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t < 128) *reinterpret_cast<f32x4*>(dst + 4 * 4096) = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
      }
      const float* tab = reinterpret_cast<const float*>(smem + CA_COEF_OFF) + c4;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(tab), sh = *reinterpret_cast<const f32x4*>(tab + 32);
      const bool own = jj >= j0 && jj < j1;                // this piece writes the by-product of its own rows only
      float* aout = p.a_out + ((img + y + p.g.ph) * Wp + px0) * 32;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        f32x4 yv = pz[k] * sc + sh;
        const f32x4 ys = yv * p.slope;                      // 0 < slope < 1: lrelu(y) = max(y, slope*y), the same bits as
        yv.x = fmaxf(yv.x, ys.x); yv.y = fmaxf(yv.y, ys.y);   // the compare-and-select of as_bn_act_fwd (host checks slope)
        yv.z = fmaxf(yv.z, ys.z); yv.w = fmaxf(yv.w, ys.w);
        if (SKIP) yv += pa[k];
        if (k == 0 || k == 4) {                            // only the halo voxels (v < 8, v >= 136) can lie outside the image
          const int xx = x0 - 8 + (t >> 3) + 32 * k;
          yv = (xx >= 0 && xx < W) ? yv : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (k < 4 || t < 128) *reinterpret_cast<f32x4*>(dst + k * 4096) = yv;
        // by-product: the segment's own 128 voxels (v in [8, 136))
        const bool mine = k == 0 ? t >= 64 : (k == 4 ? t < 64 : true);
        if (own && mine) ca_store4(aout + k * 1024, t16, yv);
      }
    };
    auto convert_row = [&](int jj) { convert_from(jj, pz, pa); };

#ifdef CA_RUNIN_SERIAL
    // (A/B builds only, EXTRA=-DCA_RUNIN_SERIAL: round 2's run-in — three dependent fetch -> wait -> convert rounds)
    fetch_row(j0 - 1); wait_row(); convert_row(j0 - 1);
    fetch_row(j0); wait_row(); convert_row(j0);
    fetch_row(j0 + 1); wait_row(); convert_row(j0 + 1);
#else
    // ---- run-in: activated rows j0-1, j0, j0+1 ----  two rows in flight at a time (two register sets: the accumulator and
    // the operand ring of the tile loop are not live yet; three sets spilled), retired by counted waits — vector-memory
    // instructions retire in order — two exposed round trips per piece instead of three
    {
      f32x4 qz[5], qa[5];
      fetch_into(j0 - 1, pz, pa); fetch_into(j0, qz, qa);
      if (SKIP) asm volatile("s_waitcnt vmcnt(10)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]) :: "memory");
      if (SKIP) asm volatile("" : "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]), "+v"(pa[4]) :: "memory");
      convert_from(j0 - 1, pz, pa);                        // never one of the piece's own rows: no by-product stores
      fetch_into(j0 + 1, pz, pa);
      if (SKIP) asm volatile("s_waitcnt vmcnt(10)" : "+v"(qz[0]), "+v"(qz[1]), "+v"(qz[2]), "+v"(qz[3]), "+v"(qz[4]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" : "+v"(qz[0]), "+v"(qz[1]), "+v"(qz[2]), "+v"(qz[3]), "+v"(qz[4]) :: "memory");
      if (SKIP) asm volatile("" : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(qa[4]) :: "memory");
      convert_from(j0, qz, qa);
      // row j0 is always an own row: exactly four by-product stores per wave (k = 1..3 everybody, k = 0 waves 1-3, k = 4
      // wave 0) queue up behind row j0+1's loads — that row is home when at most those four are outstanding
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]), "+v"(pz[4]) :: "memory");
      if (SKIP) asm volatile("" : "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]), "+v"(pa[4]) :: "memory");
      convert_from(j0 + 1, pz, pa);
    }
#endif
    __syncthreads();

    for (int j = j0; j < j1; ++j) {
      const int y = r0 + j * d;
      const char* rows[3] = {smem + ((j + 0) % 3) * CA_ROW_BYTES, smem + ((j + 1) % 3) * CA_ROW_BYTES,
                             smem + ((j + 2) % 3) * CA_ROW_BYTES};
      const int xw = x0 + 32 * wave;
#ifndef CA_EXP_NOCONV
      fetch_row(j + 2);                                    // in flight during the matrix phase
#endif
      const int vbase = 8 + 32 * wave + li;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bias_v;
      f32x4 a[4], wl[2];
      const char* wlds = smem + CA_WLDS_OFF + lane * 16;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) a[cc] = ca_chunk(rows[0], vbase - d, h, cc);
#pragma unroll
      for (int cc = 0; cc < 36; ++cc) {
        if (cc + 3 < 36) {
          const int tp = (cc + 3) >> 2;
          a[(cc + 3) & 3] = ca_chunk(rows[tp / 3], vbase + (tp % 3 - 1) * d, h, (cc + 3) & 3);
        }
        if (cc + 1 >= 4 * CA_RES && cc + 1 < 36)           // one chunk ahead: the streamed taps' B fragments
          wl[(cc + 1) & 1] = *reinterpret_cast<const f32x4*>(wlds + (cc + 1 - 4 * CA_RES) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 av = a[cc & 3];
        if (cc < 4 * CA_RES) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, R[cc >> 2][4 * (cc & 3) + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, R[cc >> 2][4 * (cc & 3) + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, R[cc >> 2][4 * (cc & 3) + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, R[cc >> 2][4 * (cc & 3) + 3], acc, 0, 0, 0);
        } else {
          const f32x4 bw = wl[cc & 1];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bw.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bw.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bw.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bw.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();                                     // B1: nobody reads row j-1's slot any more
      // ---- output: 16 stores per wave, then the tile's contribution to the moments ----
#ifndef CA_EXP_NOCONV
      wait_row();                                          // the prefetch is home (requested ~9,000 cycles ago); older
#endif
      float* z_base = p.ep.z + ((img + y + p.g.ph) * Wp + xw + p.g.pw) * 32;   // than every store below
      // The stores read the accumulator straight out of the last MFMA.  A 16-pass MFMA's result may be read by a vector-
      // memory instruction 18 wait states later at the earliest; hipcc inserts those for its own stores (s_nop 15, s_nop 1)
      // but cannot see into inline asm, and the ~17 scalar instructions in between are not a guarantee.
      asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
#define CA_ST(r) ca_store_imm<CA_ROW_IMM(r)>(z_base, io_off, acc[r]);
      CA_FOR_ROWS(CA_ST)
#undef CA_ST
#ifndef CA_EXP_NOMOM
      if (p.ep.stat_mean != nullptr) {
        const int dup = x_new - xw;      // wave-uniform: rows below `dup` also belong to the neighbouring segment
        if (dup <= 0) {
          st_c = st_n == 0.f ? acc[0] : st_c;
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float dd = acc[r] - st_c; st_s1 += dd; st_s2 = fmaf(dd, dd, st_s2); }
          st_n += 16.f;
        } else if (dup < 32) {
          st_c = st_n == 0.f ? acc[15] : st_c;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float dd = row >= dup ? acc[r] - st_c : 0.f;
            st_s1 += dd; st_s2 = fmaf(dd, dd, st_s2); st_n += row >= dup ? 1.f : 0.f;
          }
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#ifndef CA_EXP_NOCONV
      convert_row(j + 2);
#endif
      __syncthreads();                                     // B2: activated row j+2 is in place
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
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

// Applicable to the refinement geometry: 2-D 3x3 stride 1, dilation <= 8 within the halo, rows of at least 128 pixels and
// enough tiles for the fixed grid.
bool conv32_act_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (s->kd != 1 || s->kh != 3 || s->kw != 3 || s->stride != 1) return false;
  if (s->dil < 1 || s->dil > 8 || s->pad_h != s->dil || s->pad_w != s->dil) return false;
  if (gin->D != 1 || gout->D != 1 || gin->pd != 0) return false;
  if (gin->B != gout->B || gin->H != gout->H || gin->W != gout->W) return false;
  if (gin->ph != gout->ph || gin->pw != gout->pw || gin->pw < 8 || gin->ph < s->dil) return false;
  if (gout->W < 128) return false;
  const long tiles = (long)gout->B * gout->H * ((gout->W + 127) / 128);
  return tiles >= 4L * CA_GRID && tiles < (1L << 31);
}

int conv32_act_parts(void) { return ca_grid(); }

int conv32_act_launch(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift, float* a_out,
                      const as_pcl* g, const as_conv_shape* s, const float* packed_w, const float* bias, float slope,
                      float* z, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  static AsPerDevice attr_set[2];
  const int fi = a_prevprev != nullptr ? 1 : 0;
  const void* fn = fi ? reinterpret_cast<const void*>(conv32_act_kernel<true>) : reinterpret_cast<const void*>(conv32_act_kernel<false>);
  if (!attr_set[fi].get()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, CA_LDS_BYTES);
    if (e != hipSuccess) { as_set_error("as_conv32_act_fwd: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set[fi].set();
  }
  ActArgs a;
  a.zin = z_prev; a.ain = a_prevprev; a.in_scale = in_scale; a.in_shift = in_shift; a.a_out = a_out; a.wq = packed_w;
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = nullptr; a.ep.ep_shift = nullptr; a.ep.residual = nullptr;
  a.ep.stat_mean = stat_mean; a.ep.stat_m2 = stat_m2; a.ep.stat_cnt = stat_cnt; a.ep.epilogue = 0; a.ep.slope = slope;
  a.g = as_make_dev(g);
  a.dil = s->dil; a.nseg = (g->W + 127) / 128; a.slope = slope;
  void* kargs[] = {&a};
  hipError_t le = hipLaunchKernel(fn, dim3(ca_grid()), dim3(256), kargs, CA_LDS_BYTES, (hipStream_t)stream);
  if (le != hipSuccess) { as_set_error("as_conv32_act_fwd: launch failed: %s", hipGetErrorString(le)); return AS_ERR_LAUNCH; }
  return AS_OK;
}
