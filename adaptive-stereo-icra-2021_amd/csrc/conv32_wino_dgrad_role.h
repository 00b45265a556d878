// The role functions of the data gradient of a full-resolution refinement layer by minimal filtering F(2x2, 3x3) — the body of
// conv32_wino_dgrad.hip (its header comment describes the roles), shared with conv32_wino_bwd.hip, where the same four waves run
// beside four weight-gradient waves in one launch (FUSED).
#pragma once
#include "as_common.h"
#include "conv32_wino.h"
#include "conv32_wino_dev.h"

// build-time switches of the FUSED flavour (tests/tools/fused_bwd_ab.sh times every combination on one box)
#ifndef FB_PK_XF
#define FB_PK_XF 0        // the matrix loop's differences on packed instructions
#endif
#ifndef FB_PK_T
#define FB_PK_T 0         // T = (M A) on packed instructions
#endif
#ifndef FB_MASK_BRANCH
#define FB_MASK_BRANCH 0  // the shared-column mask of the next BatchNorm's sums behind a real branch
#endif
#ifndef DG_GRID
#define DG_GRID 512
#endif

// FUSED = the role functions as the data-gradient half of conv32_wino_bwd.hip (one 8-wave workgroup per CU, the weight gradient
// on waves 4-7): a ring of FIVE g_z rows (the weight-gradient waves read rows j, j+1 while rows j+3, j+4 are being converted),
// three raw slots for every dilation, no g_z by-product.
template <int L, bool FUSED = false> struct DgGeo {
  static constexpr int d = 1 << L;
  static constexpr int NV = 64 + 2 * d;                  // staged voxels per row: d + 64 + d
  static constexpr int V0 = 8 - d;                       // first staged voxel in conv32_wino.hip's 80-voxel frame (wn_addr)
  static constexpr int ROWB = NV * 128;                  // bytes of a ring row
  static constexpr int RC = 8 * NV;                      // 16-byte chunks per row
  static constexpr int K = (2 * RC + 127) / 128;         // conversion chunks per inner-wave thread and row pair: 9, 9, 9, 10
  static constexpr int RING = FUSED ? 5 : 4;             // g_z rows in LDS
  static constexpr int COEF_OFF = RING * ROWB;           // k1, k2, k3, scale, shift, mean [6][32]
  static constexpr int X_OFF = COEF_OFF + 768;           // T tiles of waves 1, 2: [2 waves][2 j][4 g][64 lanes] float4
  static constexpr int RAW_OFF = X_OFF + 16384;          // raw g_a rows [slots][64 voxels][32]
  static constexpr bool RAW_EVEN = FUSED || L < 3;       // even comb rows (wave 0's) have their two slots
  static constexpr int RAW_SLOTS = RAW_EVEN ? 3 : 1;
  static constexpr int LDS = RAW_OFF + RAW_SLOTS * 8192;
  // ring slot of comb row `row` (>= -1)
  __device__ static inline int slot(int row) { return FUSED ? (int)((unsigned)(row + 5) % 5u) : (row + 1) & 3; }
};

struct DgradArgs {
  const float* z;          // this layer's pre-activation
  const float* g_a;        // gradient w.r.t. the layer's output a = lrelu(BN(z)) + x
  float* g_z;              // by-product: gradient w.r.t. z
  float* g_x;              // dgrad(g_z) + g_a
  const float* wq;         // transformed transposed weights [16][4][64][4] (pack kind AS_PACK_WINO_T)
  const float* in_scale;   // this layer's BatchNorm affine, batch mean, stage-3 coefficients k1, k2, k3 [96]
  const float* in_shift;
  const float* bn_mean;
  const float* bn_coef;
  const float* nz;         // next BatchNorm backward (the layer below): pre-activation, affine, mean
  const float* n_scale;
  const float* n_shift;
  const float* n_mean;
  double* n_partial;       // [grid][64]: sum g_y, sum g_y * (z - mean) of the next BatchNorm
  PclDev g;
  int nseg, pairs;         // column segments per row; row pairs per (image, segment) over all combs
  float slope;
#ifdef FB_TIMING_BUILD
  long long* timing;       // diagnostic build of conv32_wino_bwd.hip only: [workgroup][wave][10] cycle counts per phase
#endif
};

// Diagnostic build (make EXTRA=-DFB_TIMING_BUILD; tests/tools/fused_bwd_timing.py prints the averages): every wave of the fused
// backward adds up the shader cycles it spends in each phase.  Slots: 0 decode + run-in, 1 matrix phase, 2 work up to B1,
// 3 wait at B1, 4 conversion / output rows / weight-gradient matrix steps, 5 wait at B2, 6 wait for loads, 7 drain.
#ifdef FB_TIMING_BUILD
#define FB_T(slot) do { if constexpr (FUSED) { const long long now_ = clock64(); tacc[slot] += now_ - tlast; tlast = now_; } } while (0)
#else
#define FB_T(slot) do { } while (0)
#endif

#define DG_FOR_8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define DG_IMM(r) (wn_c0<L>(((r) & 3) + 8 * (((r) >> 2) & 1)) * 128)
// column (in the segment) of accumulator row r of output parity jc for this lane's half h
#define DG_COL(r, jc) (wn_c0<L>(((r) & 3) + 8 * ((r) >> 2)) + wn_c0<L>(4 * h) + (jc) * d)

template <int ROLE, int L, bool FUSED = false>
__device__ __forceinline__ void dgrad_role(const DgradArgs& p, char* smem) {
  using G = DgGeo<L, FUSED>;
  constexpr int d = G::d;
  constexpr bool INNER = ROLE == 1 || ROLE == 2;
  constexpr int K = G::K;
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, li = lane & 31;
  const int H = p.g.H, W = p.g.W, Wp = p.g.Wp;
#ifdef FB_PRIO_D                                  // (experiment: static issue priority of the data-gradient waves in the one-launch backward)
  if constexpr (FUSED) __builtin_amdgcn_s_setprio(FB_PRIO_D);
#endif
#ifdef FB_TIMING_BUILD
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] long long tlast = clock64();
  [[maybe_unused]] const long long wall0 = wall_clock64();
#endif

  // the wave's four transformed filters U[ROLE][c]: R[c][4q+e] = chunk q, element e
  f32x16 R[4];
  {
    const float* wb = p.wq + (ROLE * 4) * 1024 + lane * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(wb + c * 1024 + q * 256);
        R[c][4 * q + 0] = t4.x; R[c][4 * q + 1] = t4.y; R[c][4 * q + 2] = t4.z; R[c][4 * q + 3] = t4.w;
      }
  }
  // ---- operand gather: this lane's tile li, input column m -> staged voxel 8 + c0 + (m-1) d (80-voxel frame); chunk 4h + q ----
  int op_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) op_off[m] = wn_addr<L>(8 + wn_c0<L>(li) + (m - 1) * d, 4 * h) - G::V0 * 128;
  // input rows (of the four of a tile) and sign of this wave's row transform: R = d[ra] + sg * d[rb]
  constexpr int ra = ROLE == 0 ? 0 : (ROLE == 2 ? 2 : 1);
  constexpr int rb = ROLE == 0 ? 2 : (ROLE == 1 ? 2 : (ROLE == 2 ? 1 : 3));
  constexpr bool SG_PLUS = ROLE == 1;

  // ---- inner waves: the conversion team of 128 threads; addressing: WnPair (conv32_wino_dev.h) ----
  using P = WnPair<128, L>;
  static_assert(P::K == K && P::RC == G::RC, "geometry");
  const int tid2 = (ROLE - 1) * 64 + lane;                 // 0..127 over waves 1, 2
  P pr;
  pr.init(INNER ? tid2 : lane, G::V0 * 128);
  // ---- outer waves: this wave finishes output row oi of every tile, both column parities ----
  constexpr int oi = ROLE == 3 ? 1 : 0;
  const unsigned io_off = (unsigned)(wn_c0<L>(4 * h) * 128 + 4 * li);
  float bn_sc = 0.f, bn_sh = 0.f, bn_mu = 0.f;
  float bn_dy[2] = {0.f, 0.f}, bn_dx[2] = {0.f, 0.f};       // next-BatchNorm sums per lane and column parity
  if constexpr (!INNER) { bn_sc = p.n_scale[li]; bn_sh = p.n_shift[li]; bn_mu = p.n_mean[li]; }

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
    const int x0 = min(64 * seg, W - 64);
    // one owner per column shared with the shifted last segment: conv32_wino.hip
    const int keep = (seg == p.nseg - 2 && 64 * p.nseg > W) ? W - 64 * (p.nseg - 1) : 64;
    const long img = (long)b * p.g.Hp;
    const int px0 = x0 - d + p.g.pw;                       // first staged column (padded coordinates)

    // ================= inner waves: fetch and convert a row pair (ja, ja + 1) =================
    f32x4 pz[K], pa[K];
    long pair_off = 0;                                      // float offset of row A's first staged voxel of the pair in flight
    long pair_delta = 0;                                    // row A -> row B in floats (rows outside the image are clamped: >= 0)
    const bool edge = x0 < d || x0 + 64 + d > W;            // (uniform) a halo column of this segment lies outside the image
    auto pair_rows = [&](int ja) {
      const int ya = min(max(r0 + ja * d, 0), H - 1), yb = min(max(r0 + (ja + 1) * d, 0), H - 1);
      pair_off = ((img + ya + p.g.ph) * Wp + px0) * 32;
      pair_delta = (long)(yb - ya) * Wp * 32;
    };
    auto fetch_one = [&](const float* base, f32x4 (&pv)[K]) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (P::all_a(k)) {
          wn_load4(pv[k], base + 128 * 4 * k, pr.t16);
        } else if (P::all_b(k)) {
          if (pr.active(k)) wn_load4(pv[k], base + (128 * k - P::RC) * 4 + pair_delta, pr.t16);
        } else {                                           // the chunk that straddles the two rows: threads t >= TS are in row B
          wn_load4(pv[k], base + (128 * k - P::RC) * 4, pr.t16 + (pr.t >= P::TS ? (unsigned)(pair_delta * 4) : (unsigned)(P::RC * 16)));
        }
      }
    };
    auto convert = [&](int ja, f32x4 (&pz)[K], f32x4 (&pa)[K]) {           // -> the ring slots of rows ja, ja + 1
      const float* tab = reinterpret_cast<const float*>(smem + G::COEF_OFF) + (tid2 & 7) * 4;
      const f32x4 k1 = *reinterpret_cast<const f32x4*>(tab), k2 = *reinterpret_cast<const f32x4*>(tab + 32);
      const f32x4 k3 = *reinterpret_cast<const f32x4*>(tab + 64), sc = *reinterpret_cast<const f32x4*>(tab + 96);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(tab + 128), bmu = *reinterpret_cast<const f32x4*>(tab + 160);
      // per row of the pair (uniform): inside the image?  one of this piece's own rows?  ring slot, raw slot, by-product row
      const int y_a = r0 + ja * d, y_b = r0 + (ja + 1) * d;
      const bool in_a = y_a >= 0 && y_a < H, in_b = y_b >= 0 && y_b < H;
      const bool own_a = ja >= j0 && ja < j1, own_b = ja + 1 >= j0 && ja + 1 < j1;
      int ring_a_off = G::slot(ja) * G::ROWB, ring_b_off = G::slot(ja + 1) * G::ROWB;
      if constexpr (FUSED) asm volatile("" : "+s"(ring_a_off), "+s"(ring_b_off));   // (opaque: see the tile loop)
      char* ring_a = smem + ring_a_off;
      char* ring_b = smem + ring_b_off;
      // raw g_a row for the skip connection: odd comb rows slot 0, even rows slot 1 + (row / 2) % 2; voxel vq - d of the slot
      auto raw_slot = [&](int jj) { return (jj & 1) ? 0 : 1 + ((jj >> 1) & 1); };
      const bool raw_a = own_a && (G::RAW_EVEN || (ja & 1)), raw_b = own_b && (G::RAW_EVEN || ((ja + 1) & 1));
      char* rawp_a = smem + G::RAW_OFF + raw_slot(ja) * 8192 - d * 128;
      char* rawp_b = smem + G::RAW_OFF + raw_slot(ja + 1) * 8192 - d * 128 - G::NV * 128;
      [[maybe_unused]] float* out_a = FUSED ? nullptr : p.g_z + pair_off;   // (fused: g_z stays on the chip)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool strad = !P::all_a(k) && !P::all_b(k);
        const bool rowb = pr.in_b(k);                       // (per lane only for the straddling chunk)
        const bool act = pr.active(k);
        // stage 3 of the BatchNorm backward (conv32_wino.hip MODE 2 / conv32_bwd.hip's arithmetic)
        const f32x4 ga = pa[k], zz = pz[k];
        const f32x4 yy = zz * sc + sh;
        const f32x4 gl = ga * p.slope;
        f32x4 gy;
        gy.x = yy.x > 0.f ? ga.x : gl.x; gy.y = yy.y > 0.f ? ga.y : gl.y;
        gy.z = yy.z > 0.f ? ga.z : gl.z; gy.w = yy.w > 0.f ? ga.w : gl.w;
        f32x4 yv = (gy - k1 - (zz - bmu) * k2) * k3;
        // halo voxels: only they can lie outside the image (zero padding), and they have no by-product and no raw copy; which
        // chunks can hold any is known at compile time
        bool halo = false;
        if (P::may_halo(k, false) || P::may_halo(k, true)) {
          const int vq = pr.vq(k, rowb);
          halo = vq < d || vq >= 64 + d;
          if (edge && halo) {
            const int xx = x0 - d + vq;
            if (!(xx >= 0 && xx < W)) yv = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
        if (!(rowb ? in_b : in_a)) yv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (act) *reinterpret_cast<f32x4*>((rowb ? ring_b : ring_a) + (strad ? (rowb ? pr.lds_b(k) : pr.lds_a(k)) : (P::all_b(k) ? pr.lds_b(k) : pr.lds_a(k)))) = yv;
        if (act && (rowb ? own_b : own_a) && !halo) {
          if constexpr (!FUSED) {
            if (P::all_a(k)) wn_store4(out_a + 128 * 4 * k, pr.t16, yv);
            else if (P::all_b(k)) wn_store4(out_a + (128 * k - P::RC) * 4 + pair_delta, pr.t16, yv);
            else wn_store4(out_a + (128 * k - P::RC) * 4, pr.t16 + (rowb ? (unsigned)(pair_delta * 4) : (unsigned)(P::RC * 16)), yv);
          }
          // (the raw row's voxel is (t >> 3) + 16 k [- NV] - d: linear in t as well)
          if (rowb ? raw_b : raw_a) *reinterpret_cast<f32x4*>((rowb ? rawp_b : rawp_a) + 16 * k * 128 + pr.t16) = ga;
        }
        if (k & 1) __builtin_amdgcn_sched_barrier(0);      // (two chunks at a time: bounded register appetite)
      }
    };
    auto wait_all = [&](f32x4 (&pv)[K]) {                   // every load so far is home (in-order retirement)
      if constexpr (K == 9)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]),
                     "+v"(pv[6]), "+v"(pv[7]), "+v"(pv[8]) :: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]),
                     "+v"(pv[6]), "+v"(pv[7]), "+v"(pv[8]), "+v"(pv[K - 1]) :: "memory");
    };
    auto touch = [&](f32x4 (&pv)[K]) {                      // (no instruction: ties the registers to the wait above)
      if constexpr (K == 9)
        asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]),
                     "+v"(pv[8]) :: "memory");
      else
        asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]),
                     "+v"(pv[8]), "+v"(pv[K - 1]) :: "memory");
    };

    // ---- run-in: g_z rows j0-1 .. j0+2 (inner waves, one pair after the other; the outer waves wait at the barrier) ----
    if constexpr (INNER) {
#pragma unroll
      for (int k = 0; k < K; ++k) { pz[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; pa[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      pair_rows(j0 - 1);
      fetch_one(p.z + pair_off, pz); fetch_one(p.g_a + pair_off, pa);
      wait_all(pz); touch(pa);
      convert(j0 - 1, pz, pa);
      pair_rows(j0 + 1);
      fetch_one(p.z + pair_off, pz); fetch_one(p.g_a + pair_off, pa);
      wait_all(pz); touch(pa);
      convert(j0 + 1, pz, pa);
    }
    __syncthreads();
    FB_T(0);

    for (int j = j0; j < j1; j += 2) {
      const bool more = j + 2 < j1;                        // (the last tile of a piece converts nothing)
      if constexpr (INNER) {
        if (more) { pair_rows(j + 3); fetch_one(p.z + pair_off, pz); }  // in flight during the matrix phase (g_a follows it)
      }
      // input row m of the tile = comb row j-1+m.  FUSED: the slot is (row mod 5); left visible, hipcc turns "slot * ROWB + this
      // lane's 32 operand offsets" into 32 loop-carried vector registers, spills them and reloads them ONE AFTER THE OTHER at
      // the head of every tile (12,000 cycles per tile, measured with -DFB_TIMING_BUILD) — so the scalar is made opaque
      int row_a_off = G::slot(j - 1 + ra) * G::ROWB, row_b_off = G::slot(j - 1 + rb) * G::ROWB;
      if constexpr (FUSED) asm volatile("" : "+s"(row_a_off), "+s"(row_b_off));
      const char* row_a = smem + row_a_off;
      const char* row_b = smem + row_b_off;
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
          constexpr bool PKX = FUSED && FB_PK_XF;
#pragma unroll
          for (int m = 0; m < 4; ++m) Rt[m] = SG_PLUS ? xa[m] + xb[m] : wn_sub4_if<PKX>(xa[m], xb[m]);
          // (FUSED: the differences on packed instructions — conv32_wino_dev.h; a vector instruction costs the SIMD ~5 cycles of
          //  matrix time whichever of its two waves issues it, and here nothing else hides them)
          V[0] = wn_sub4_if<PKX>(Rt[0], Rt[2]); V[1] = Rt[1] + Rt[2]; V[2] = wn_sub4_if<PKX>(Rt[2], Rt[1]);
          V[3] = wn_sub4_if<PKX>(Rt[1], Rt[3]);
          if constexpr (PKX) wn_before_mfma();
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
      // T[jc] = (M[r] A)[jc]: A^T = [1 1 1 0; 0 1 -1 -1]
      f32x16 T0, T1;
      if constexpr (FUSED && FB_PK_T) {
        wn_after_mfma();                                   // (the matrix phase ended at a sched_barrier right above)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          f32x4 a4[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) a4[c] = (f32x4){acc[c][4 * gq + 0], acc[c][4 * gq + 1], acc[c][4 * gq + 2], acc[c][4 * gq + 3]};
          const f32x4 t0 = (a4[0] + a4[1]) + a4[2];
          const f32x4 t1 = wn_sub4(wn_sub4(a4[1], a4[2]), a4[3]);
          T0[4 * gq + 0] = t0.x; T0[4 * gq + 1] = t0.y; T0[4 * gq + 2] = t0.z; T0[4 * gq + 3] = t0.w;
          T1[4 * gq + 0] = t1.x; T1[4 * gq + 1] = t1.y; T1[4 * gq + 2] = t1.z; T1[4 * gq + 3] = t1.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          T0[r] = (acc[0][r] + acc[1][r]) + acc[2][r];
          T1[r] = (acc[1][r] - acc[2][r]) - acc[3][r];
        }
      }
      __builtin_amdgcn_sched_barrier(0);                   // (the accumulators die HERE, before anything below asks for registers)
      FB_T(1);

      if constexpr (INNER) {
        // ---------------- publish T[0], T[1]; request the g_a rows of the pair; B1; convert; B2 ----------------
        char* xw = smem + G::X_OFF + ((ROLE - 1) * 2) * 4096 + lane * 16;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          *reinterpret_cast<f32x4*>(xw + gq * 1024) = (f32x4){T0[4 * gq], T0[4 * gq + 1], T0[4 * gq + 2], T0[4 * gq + 3]};
          *reinterpret_cast<f32x4*>(xw + 4096 + gq * 1024) = (f32x4){T1[4 * gq], T1[4 * gq + 1], T1[4 * gq + 2], T1[4 * gq + 3]};
        }
        if (more) fetch_one(p.g_a + pair_off, pa);                  // (the accumulators are dead: registers to spare)
        FB_T(2);
        __syncthreads();                                   // B1: the T tiles are in place; nobody reads rows j-1, j any more
        FB_T(3);
        if (more) {
          wait_all(pz); touch(pa);
          FB_T(6);
          convert(j + 3, pz, pa);
        }
        FB_T(4);
        __syncthreads();                                   // B2: g_z rows j+3, j+4 and their raw rows are in place
        FB_T(5);
      } else {
        // ---------------- outer: skip rows and next pre-activation requested; B1; finish the output row; B2 ----------------
        const int yrow = j + oi;
        const bool row_ok = yrow < j1;                     // (wave-uniform) the pair's second row may lie outside the piece
        float res[2][16], zt[2][16];
        const long ovox = ((img + r0 + yrow * d + p.g.ph) * Wp + x0 + p.g.pw) * 32;
        if (row_ok) {
#define DG_LDZ(r) wn_load_imm<DG_IMM(r)>(zt[0][r], p.nz + ovox, io_off); wn_load_imm<DG_IMM(r)>(zt[0][8 + r], p.nz + ovox, io_off + 4096u); \
                  wn_load_imm<DG_IMM(r) + d * 128>(zt[1][r], p.nz + ovox, io_off); wn_load_imm<DG_IMM(r) + d * 128>(zt[1][8 + r], p.nz + ovox, io_off + 4096u);
          DG_FOR_8(DG_LDZ)
#undef DG_LDZ
          if constexpr (oi == 1 || G::RAW_EVEN) {
            // the raw g_a row the inner waves parked when they converted it (odd comb rows: slot 0; even: 1 + (row / 2) % 2)
            const int slot = oi == 1 ? 0 : 1 + ((j >> 1) & 1);
            const char* rr = smem + G::RAW_OFF + slot * 8192 + io_off;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int imm = DG_IMM(r & 7) + (r >> 3) * 4096;
              res[0][r] = *reinterpret_cast<const float*>(rr + imm);
              res[1][r] = *reinterpret_cast<const float*>(rr + imm + d * 128);
            }
          } else {
#define DG_LDA(r) wn_load_imm<DG_IMM(r)>(res[0][r], p.g_a + ovox, io_off); wn_load_imm<DG_IMM(r)>(res[0][8 + r], p.g_a + ovox, io_off + 4096u); \
                  wn_load_imm<DG_IMM(r) + d * 128>(res[1][r], p.g_a + ovox, io_off); wn_load_imm<DG_IMM(r) + d * 128>(res[1][8 + r], p.g_a + ovox, io_off + 4096u);
            DG_FOR_8(DG_LDA)
#undef DG_LDA
          }
        }
        int keep_l = keep;                                 // (opaque: hipcc otherwise hoists the shared-column masks of the
        asm volatile("" : "+s"(keep_l));                   //  sums out of the tile loop, into registers)
        FB_T(2);
        __syncthreads();                                   // B1
        FB_T(3);
        // Y[oi][jc] = (A^T T)[oi][jc]: (T0 + T1) + T2 for the first output row, (T1 - T2) - T3 for the second
        f32x16 Y[2];
        {
          const char* xr = smem + G::X_OFF + lane * 16;
#pragma unroll
          for (int jc = 0; jc < 2; ++jc)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const f32x4 u1 = *reinterpret_cast<const f32x4*>(xr + jc * 4096 + gq * 1024);             // T of wave 1
              const f32x4 u2 = *reinterpret_cast<const f32x4*>(xr + (2 + jc) * 4096 + gq * 1024);       // T of wave 2
              const f32x16& own = jc == 0 ? T0 : T1;
              const f32x4 o4 = (f32x4){own[4 * gq], own[4 * gq + 1], own[4 * gq + 2], own[4 * gq + 3]};
              const f32x4 yv = oi == 0 ? (o4 + u1) + u2 : (u1 - u2) - o4;
              Y[jc][4 * gq + 0] = yv.x; Y[jc][4 * gq + 1] = yv.y; Y[jc][4 * gq + 2] = yv.z; Y[jc][4 * gq + 3] = yv.w;
            }
        }
        if (row_ok) {
          // every load of this tile is home (next pre-activation, and g_a where it still comes from HBM); the stores are younger
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(zt[0][0]), "+v"(zt[0][1]), "+v"(zt[0][2]), "+v"(zt[0][3]), "+v"(zt[0][4]), "+v"(zt[0][5]), "+v"(zt[0][6]),
                         "+v"(zt[0][7]), "+v"(zt[0][8]), "+v"(zt[0][9]), "+v"(zt[0][10]), "+v"(zt[0][11]), "+v"(zt[0][12]),
                         "+v"(zt[0][13]), "+v"(zt[0][14]), "+v"(zt[0][15]) :: "memory");
          asm volatile("" : "+v"(zt[1][0]), "+v"(zt[1][1]), "+v"(zt[1][2]), "+v"(zt[1][3]), "+v"(zt[1][4]), "+v"(zt[1][5]), "+v"(zt[1][6]),
                            "+v"(zt[1][7]), "+v"(zt[1][8]), "+v"(zt[1][9]), "+v"(zt[1][10]), "+v"(zt[1][11]), "+v"(zt[1][12]),
                            "+v"(zt[1][13]), "+v"(zt[1][14]), "+v"(zt[1][15]) :: "memory");
          if constexpr (!(oi == 1 || G::RAW_EVEN)) {
            asm volatile("" : "+v"(res[0][0]), "+v"(res[0][1]), "+v"(res[0][2]), "+v"(res[0][3]), "+v"(res[0][4]), "+v"(res[0][5]),
                              "+v"(res[0][6]), "+v"(res[0][7]), "+v"(res[0][8]), "+v"(res[0][9]), "+v"(res[0][10]), "+v"(res[0][11]),
                              "+v"(res[0][12]), "+v"(res[0][13]), "+v"(res[0][14]), "+v"(res[0][15]) :: "memory");
            asm volatile("" : "+v"(res[1][0]), "+v"(res[1][1]), "+v"(res[1][2]), "+v"(res[1][3]), "+v"(res[1][4]), "+v"(res[1][5]),
                              "+v"(res[1][6]), "+v"(res[1][7]), "+v"(res[1][8]), "+v"(res[1][9]), "+v"(res[1][10]), "+v"(res[1][11]),
                              "+v"(res[1][12]), "+v"(res[1][13]), "+v"(res[1][14]), "+v"(res[1][15]) :: "memory");
          }
          const int y = r0 + yrow * d;
          float* gx_base = p.g_x + ((img + y + p.g.ph) * Wp + x0 + p.g.pw) * 32;
          const auto rs = __builtin_amdgcn_make_buffer_rsrc(gx_base, 0, keep_l * 128, 0x00020000);
#define DG_ST(r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Y[jc][r]), rs, (int)io_off + jc * d * 128 + DG_IMM(r), 0, 0); \
                 __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Y[jc][8 + r]), rs, (int)io_off + 4096 + jc * d * 128 + DG_IMM(r), 0, 0);
          if constexpr (FUSED && FB_MASK_BRANCH) {
            // The shared-column mask of the sums behind a REAL branch (one segment of twenty shares columns; as selects the mask is
            // 64 vector instructions per tile).  The branch stands BEFORE the stores: with stores in flight hipcc drains the
            // whole queue at the join (measured: +25-40 % on the launch).
#pragma unroll
            for (int jc = 0; jc < 2; ++jc)
#pragma unroll
              for (int r = 0; r < 16; ++r) Y[jc][r] += res[jc][r];
            if (keep_l == 64) {
              asm volatile("" ::: "memory");
#pragma unroll
              for (int jc = 0; jc < 2; ++jc)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const float yv = fmaf(zt[jc][r], bn_sc, bn_sh);
                  const float gy = yv > 0.f ? Y[jc][r] : Y[jc][r] * p.slope;
                  bn_dy[jc] += gy; bn_dx[jc] = fmaf(gy, zt[jc][r] - bn_mu, bn_dx[jc]);
                }
            } else {
#pragma unroll
              for (int jc = 0; jc < 2; ++jc)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const float yv = fmaf(zt[jc][r], bn_sc, bn_sh);
                  float gy = yv > 0.f ? Y[jc][r] : Y[jc][r] * p.slope;
                  gy = DG_COL(r, jc) < keep_l ? gy : 0.f;
                  bn_dy[jc] += gy; bn_dx[jc] = fmaf(gy, zt[jc][r] - bn_mu, bn_dx[jc]);
                }
            }
#pragma unroll
            for (int jc = 0; jc < 2; ++jc) { DG_FOR_8(DG_ST) }
          } else {
#pragma unroll
            for (int jc = 0; jc < 2; ++jc) {
#pragma unroll
              for (int r = 0; r < 16; ++r) Y[jc][r] += res[jc][r];
              DG_FOR_8(DG_ST)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float yv = fmaf(zt[jc][r], bn_sc, bn_sh);
                float gy = yv > 0.f ? Y[jc][r] : Y[jc][r] * p.slope;
                if (keep_l < 64) gy = DG_COL(r, jc) < keep_l ? gy : 0.f;
                bn_dy[jc] += gy; bn_dx[jc] = fmaf(gy, zt[jc][r] - bn_mu, bn_dx[jc]);
              }
            }
          }
#undef DG_ST
        }
        __builtin_amdgcn_sched_barrier(0);
        FB_T(4);
        __syncthreads();                                   // B2
        FB_T(5);
      }
    }
  }
#ifdef FB_TIMING_BUILD
  if constexpr (FUSED) {
    FB_T(7);
    if (p.timing && lane == 0) {
      long long* o = p.timing + ((long)blockIdx.x * 8 + ROLE) * 10;
      for (int i = 0; i < 8; ++i) o[i] = tacc[i];
      o[8] = wall_clock64() - wall0;
      o[9] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID: wave [3:0], SIMD [5:4], CU [11:8], SE [15:13]
    }
  }
#endif

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // next-BatchNorm sums: 8 (output position, half) partials per channel -> one fp64 pair per workgroup, in the order of the
  // first generation (its wave = oi * 2 + jc)
  float* scr = reinterpret_cast<float*>(smem);            // [8][2][32]
  if constexpr (!INNER) {
#pragma unroll
    for (int jc = 0; jc < 2; ++jc) {
      scr[(((oi * 2 + jc) * 2 + h) * 2 + 0) * 32 + li] = bn_dy[jc];
      scr[(((oi * 2 + jc) * 2 + h) * 2 + 1) * 32 + li] = bn_dx[jc];
    }
  }
  __syncthreads();
  if constexpr (ROLE == 0) {
    const int which = lane >> 5, cch = lane & 31;
    double sum = 0.0;
    for (int q = 0; q < 8; ++q) sum += (double)scr[(q * 2 + which) * 32 + cch];
    p.n_partial[(long)blockIdx.x * 64 + which * 32 + cch] = sum;
  }
}

