// kernels_chain.h -- the FTRL (n, z) update of VERY HOT features: those that occur in more than
// kHugeMin rows of the block (FFM::update_vector_nz src/model/ffm.cpp:90-136 incl. :118, and
// FM::update_vector_nz src/model/fm.cpp:80-101, applied to one element by hundreds of rows).
//
// A hot feature's touches form one sequential chain per element: n_t = n_{t-1} + g_t^2,
// z_t = (z_{t-1} + g_t) - sigma_t * w, with sigma_t a function of n_{t-1}.  Only those two
// recurrences are serial; the gradients, both square roots and the alpha divide of a touch depend
// on n_{t-1} alone.  So the chain is laid across the lanes of a DPP row: lane = (element, touch),
// 16 consecutive touches of 4 elements per wave step.  Per step:
//   (1) parallel: gradient g, g^2 and the :118 product for all 16 touches;
//   (2) serial:   running n by fifteen in-place `v_add_f32_dpp row_shr:1` steps -- lane t adds its
//                 increment to its left neighbour's running value; lane 0 has no left neighbour in
//                 its row, so the hardware leaves it alone (bound_ctrl off): it keeps
//                 "carry + own increment".  After step r lanes 0..r hold the exact sequential
//                 prefix, each addition rounded as the one-thread loop rounds it
//                 (tools/dpp_probe.hip checks the bits against a scalar loop);
//   (3) parallel: every touch's sigma * w from its n-before (the left neighbour's running n);
//   (4) serial:   running z the same way, one fused DPP add and one subtract per touch;
//   the carries into the next step come back to lane 0 with one row_mirror move.
// A wave owns one whole slot (k = 16: four such chains, one per group of 4 factors) and issues the
// four independent chains interleaved, so the DPP read-after-write wait states are filled with the
// other chains' adds instead of s_nop.  Against the previous shape (4 touches per step, quad DPP)
// the serial part per touch is the same three instructions, but the parallel part, the loads and
// the wave votes are amortised over four times as many touches, and a chain of c touches takes
// c/16 step latencies instead of c/4: the 1000-touch chains that used to set the update phase's
// span (~430 us) finish in a few tens of microseconds.
#pragma once
#include "engine_types.h"

namespace ftrl_dev {

constexpr int kChainT = 16;  // touches per step = lanes of a DPP row

#define FTRL_DPP_SHR " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define FTRL_REP15(x) x x x x x x x x x x x x x x x

// G independent running sums, one per register, over the 16 lanes of every DPP row:
// S[g] (lane t) <- S[g] (lane t-1) + q[g] (lane t), fifteen times; lane 0 of a row is never written.
template <int G>
__device__ __forceinline__ void row_chain_add(float (&S)[G], const float (&q)[G]);
template <>
__device__ __forceinline__ void row_chain_add<4>(float (&S)[4], const float (&q)[4]) {
  asm volatile("s_nop 1\n\t" FTRL_REP15(
                   "v_add_f32_dpp %0, %0, %4" FTRL_DPP_SHR "v_add_f32_dpp %1, %1, %5" FTRL_DPP_SHR
                   "v_add_f32_dpp %2, %2, %6" FTRL_DPP_SHR "v_add_f32_dpp %3, %3, %7" FTRL_DPP_SHR)
               : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3])
               : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
}
template <>
__device__ __forceinline__ void row_chain_add<2>(float (&S)[2], const float (&q)[2]) {
  asm volatile("s_nop 1\n\t" FTRL_REP15("v_add_f32_dpp %0, %0, %2" FTRL_DPP_SHR
                                        "v_add_f32_dpp %1, %1, %3" FTRL_DPP_SHR "s_nop 0\n\t")
               : "+v"(S[0]), "+v"(S[1])
               : "v"(q[0]), "v"(q[1]));
}
template <>
__device__ __forceinline__ void row_chain_add<1>(float (&S)[1], const float (&q)[1]) {
  asm volatile("s_nop 1\n\t" FTRL_REP15("v_add_f32_dpp %0, %0, %1" FTRL_DPP_SHR "s_nop 1\n\t")
               : "+v"(S[0])
               : "v"(q[0]));
}

// The z recurrence: Z (lane t) <- (Z (lane t-1) + ga (lane t)) - mc (lane t).  Lane 0 of a row skips
// the add (no source lane) but not the subtract: callers pass mc = +0.0f there (x - +0.0f == x).
template <int G>
__device__ __forceinline__ void row_chain_addsub(float (&Z)[G], const float (&ga)[G],
                                                 const float (&mc)[G]);
template <>
__device__ __forceinline__ void row_chain_addsub<4>(float (&Z)[4], const float (&ga)[4],
                                                    const float (&mc)[4]) {
  asm volatile("s_nop 1\n\t" FTRL_REP15(
                   "v_add_f32_dpp %0, %0, %4" FTRL_DPP_SHR "v_add_f32_dpp %1, %1, %5" FTRL_DPP_SHR
                   "v_add_f32_dpp %2, %2, %6" FTRL_DPP_SHR "v_add_f32_dpp %3, %3, %7" FTRL_DPP_SHR
                   "v_sub_f32 %0, %0, %8\n\tv_sub_f32 %1, %1, %9\n\t"
                   "v_sub_f32 %2, %2, %10\n\tv_sub_f32 %3, %3, %11\n\t")
               : "+v"(Z[0]), "+v"(Z[1]), "+v"(Z[2]), "+v"(Z[3])
               : "v"(ga[0]), "v"(ga[1]), "v"(ga[2]), "v"(ga[3]), "v"(mc[0]), "v"(mc[1]), "v"(mc[2]),
                 "v"(mc[3]));
}
template <>
__device__ __forceinline__ void row_chain_addsub<2>(float (&Z)[2], const float (&ga)[2],
                                                    const float (&mc)[2]) {
  asm volatile("s_nop 1\n\t" FTRL_REP15("v_add_f32_dpp %0, %0, %2" FTRL_DPP_SHR
                                        "v_add_f32_dpp %1, %1, %3" FTRL_DPP_SHR
                                        "v_sub_f32 %0, %0, %4\n\tv_sub_f32 %1, %1, %5\n\t")
               : "+v"(Z[0]), "+v"(Z[1])
               : "v"(ga[0]), "v"(ga[1]), "v"(mc[0]), "v"(mc[1]));
}
template <>
__device__ __forceinline__ void row_chain_addsub<1>(float (&Z)[1], const float (&ga)[1],
                                                    const float (&mc)[1]) {
  asm volatile("s_nop 1\n\t" FTRL_REP15("v_add_f32_dpp %0, %0, %1" FTRL_DPP_SHR
                                        "v_sub_f32 %0, %0, %2\n\ts_nop 1\n\t")
               : "+v"(Z[0])
               : "v"(ga[0]), "v"(mc[0]));
}
#undef FTRL_REP15
#undef FTRL_DPP_SHR

// every lane gets its left neighbour's value inside its DPP row; lane 0 of the row gets `first`
__device__ __forceinline__ float row_left(float first, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v),
                                                    0x111 /* row_shr:1 */, 0xf, 0xf, false));
}
// lane t of a row gets lane 15-t: what lane 0 reads is the row's last lane (the step's carry-out)
__device__ __forceinline__ float row_mirror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140 /* row_mirror */,
                                                    0xf, 0xf, false));
}
__device__ __forceinline__ float row_first(float v) {  // lane 0 of my row, to every lane of the row
  return __shfl(v, threadIdx.x & 48, 64);
}

// One step's worth of (3): sigma * w of every touch from its n-before, in the short exact forms
// when one wave vote says every operand is comfortably normal (ftrl_math.h), else IEEE.
template <int G>
__device__ __forceinline__ void chain_sigma_w(const Hyper &h, const float (&nb)[G],
                                              const float (&arg0)[G], const bool (&simple)[G],
                                              const float (&w)[G], float (&mm)[G]) {
  bool ok = h.fast_div != 0;
#pragma unroll
  for (int g = 0; g < G; g++) ok = ok && chain_operand_ok(arg0[g]) && chain_operand_ok(nb[g]);
  if (__all(ok)) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      const float d = sqrt_fast(arg0[g]) - sqrt_fast(nb[g]);
      mm[g] = div_alpha_fast(h, simple[g] ? d : 0.0f) * w[g];
    }
  } else {
#pragma unroll
    for (int g = 0; g < G; g++) {
      const float d = sqrtf(arg0[g]) - sqrtf(nb[g]);
      mm[g] = ((simple[g] ? d : 0.0f) / h.alpha) * w[g];
    }
  }
}

// The same when every live touch of the step puts its own g*g under the root (FFM touches whose own
// entry is the pair's first, every FM touch): sqrt(n + g*g) of touch t IS sqrt(n-before) of touch
// t+1, so one square root per touch serves both -- the second comes from the left neighbour by the
// DPP move that already brings its running n (lane 0: from the previous step, through sqc).
// S = running n after each touch; nc_in = the carry the step started from (lane 0 of a row).
// Returns false (nothing written) when an operand is outside the short forms' range.
template <int G>
__device__ __forceinline__ bool chain_sigma_w_forward(const Hyper &h, const float (&S)[G],
                                                      const float (&nc_in)[G], const bool (&simple)[G],
                                                      const float (&w)[G], float (&sqc)[G],
                                                      bool sq_valid, float (&mm)[G]) {
  bool ok = h.fast_div != 0;
#pragma unroll
  for (int g = 0; g < G; g++) ok = ok && chain_operand_ok(S[g]) && (sq_valid || chain_operand_ok(nc_in[g]));
  if (!__all(ok)) return false;
  if (!sq_valid) {
    // (a real branch: as a select the compiler evaluates this root in every step)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g = 0; g < G; g++) sqc[g] = sqrt_fast(nc_in[g]);
  }
#pragma unroll
  for (int g = 0; g < G; g++) {
    const float sq = sqrt_fast(S[g]);
    const float d = sq - row_left(sqc[g], sq);
    mm[g] = div_alpha_fast(h, simple[g] ? d : 0.0f) * w[g];
    sqc[g] = row_mirror(sq);  // lane 0: sqrt of the carry the next step starts from
  }
  return true;
}

// ---- FFM ------------------------------------------------------------------------------------
// Work item = (very hot feature, slot of its record, pass over G groups of 4 factors of the slot);
// one wave per item.  Lane: tl = touch inside the step (DPP row position), el = factor inside a
// group.  Touch facts come as the occurrence-ordered streams the row kernel wrote (s.haux, s.hmeta).
//
// A chain of c touches is c/16 dependent steps, and a step's inputs are gathers (the fact of every
// touch, then -- at the address the fact carries -- the partner's weights): ~2 us of latency each
// when the chip is busy.  With the loads one step ahead a step cost that latency (measured 2.4 us
// per step for the 8600-touch chains of an 8-GPU job's 65536-row blocks: one wave per slot, alone
// on the chip, 1.3 ms).  So the loop runs in chunks of kChainChunk steps over three stages: the
// raw facts of chunk c+2 are in flight while the partner weights of chunk c+1 are requested (from
// the facts that arrived a chunk ago) and chunk c is computed -- eight steps of latency tolerance
// for the facts, four for the weights, in straight-line code so that the loads retire in order
// behind one counted wait.  Slots with a multi-valued partner field somewhere in the block
// (s.cmask) take the plain one-step-ahead loop with the sequential fallback inside.
constexpr int kChainChunk = 1;  // steps per pipeline stage (2 / 4 measured slower: DESIGN.md section 7)

struct ChainTouch {  // what a step needs of one touch once its partner weights are requested
  float tg, xm, xo;
  int fl;
};

// One fast step: all 16 touches plain (no multi-valued field among them).
template <int G>
__device__ __forceinline__ void chain_step(const ModelDev &m, unsigned long long own_bits, bool in_range, bool l0,
                                           const ChainTouch &f, const float (&vp)[G],
                                           const bool (&act)[G], const float (&w)[G],
                                           float (&nc)[G], float (&zc)[G], float (&sqc)[G],
                                           bool &sq_valid) {
  const int fl = f.fl;
  const bool smp = in_range && owns_bit(own_bits, fl >> 8) && (fl & HF_SIMPLE) != 0;
  const bool first = (fl & HF_FIRST) || m.h.learn;
  const float tg = f.tg;
  const float x = f.xm * f.xo;  // x_own*x_other or x_other*x_own: same product
  float g1v[G], q[G], S[G], ga[G], arg0[G], nb[G], mm[G], mc[G], Z[G], nc_in[G];
  bool simple[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    nc_in[g] = nc[g];
    simple[g] = smp && act[g];
    const float gr = tg * vp[g] * x;  // own slot's gradient (g1 if own entry first, else g2)
    const float g1 = tg * w[g] * x;   // second-entry case: the first entry's gradient
    const float gg = gr * gr;
    g1v[g] = first ? gg : gr * g1;    // what the square root sees added to n (ffm.cpp:113 / :118)
    ga[g] = simple[g] ? gr : -0.0f;
    q[g] = simple[g] ? gg : -0.0f;    // x + -0.0f == x bit for bit: idle touches apply nothing
    S[g] = nc[g] + q[g];              // lane 0: n after its touch
  }
  row_chain_add<G>(S, q);
#pragma unroll
  for (int g = 0; g < G; g++) nc[g] = row_mirror(S[g]);  // lane 0: the row's last running n
  // all live touches on the g*g side: one square root per touch (see chain_sigma_w_forward)
  const bool fwd = __all(first || !smp);
  if (!fwd || !chain_sigma_w_forward<G>(m.h, S, nc_in, simple, w, sqc, sq_valid, mm)) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      nb[g] = row_left(nc_in[g], S[g]);  // n before this touch
      arg0[g] = nb[g] + g1v[g];
    }
    chain_sigma_w<G>(m.h, nb, arg0, simple, w, mm);
    sq_valid = false;
  } else {
    sq_valid = true;
  }
#pragma unroll
  for (int g = 0; g < G; g++) {
    const float ms = simple[g] ? mm[g] : 0.0f;
    mc[g] = l0 ? 0.0f : ms;
    Z[g] = (zc[g] + ga[g]) - ms;      // lane 0: z after its touch
  }
  row_chain_addsub<G>(Z, ga, mc);
#pragma unroll
  for (int g = 0; g < G; g++) zc[g] = row_mirror(Z[g]);
}

// The items of one list of features (s.huge or s.giant), G interleaved chains per wave; `wave` of
// `n_waves` waves share the list.
// c_lo <= occurrences < c_hi: the share of the list this call walks
template <int G>
__device__ __forceinline__ void ffm_chain_items(const ModelDev &m, const Rows &rows, const Scratch &s,
                                                const int *list, int n_list, unsigned wave,
                                                unsigned n_waves, int ph, int phases, int c_lo = 0,
                                                int c_hi = 0x7fffffff) {
  // One chain per wave (the giant features) takes 0.9 us per 16-touch step even alone on the
  // chip, three times its issue cost; deeper prefetch for it (chunks of 2, 4, 8 steps: facts and
  // weights up to 16 / 8 steps ahead, 107 VGPRs) measured 0 to 5 % SLOWER per step on an 8-GPU
  // rank's blocks, a launch of its own at twice the occupancy (FFM_GIANT_APART) 5 % slower, issue
  // priority (FFM_GIANT_PRIO) the same: kept as knobs, off.
  constexpr int CH = kChainChunk;
  const int RL = m.row_len, k = m.n_factors, F = m.n_fields;
  const int lane = threadIdx.x & 63;
  const int tl = lane & (kChainT - 1), el = lane >> 4;
  const bool l0 = tl == 0;
  const int groups = k >> 2;                    // groups of 4 factors per slot
  const int passes = (groups + G - 1) / G;      // 1 for k <= 16 at G = 4
  const int slots = record_span(m, 1);          // slots walked per record
  const unsigned per_feat = static_cast<unsigned>(slots) * passes;
  const unsigned n_items = static_cast<unsigned>(n_list) * per_feat;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned li = item / per_feat;
    const int rem = static_cast<int>(item - li * per_feat);
    const int sc = rem / passes, pass = rem - sc * passes;
    const int u = wave_uniform(list[li]);
    const int4 ud = s.udesc[u];  // {feature, start, count, field}
    const int fa = wave_uniform(ud.w);
    const int fp = wave_uniform(walk_field(m, fa, sc));  // partner field of slot sc
    if (fp < 0) continue;
    const int start = wave_uniform(ud.y);
    const int c_all = wave_uniform(ud.z);
    if (c_all < c_lo || c_all >= c_hi) continue;  // the other call's
    int t_lo, c;  // this row phase's touches [t_lo, c) of the feature's occurrences
    phase_touches(s, start, c_all, ph, phases, t_lo, c);
    t_lo = wave_uniform(t_lo);
    c = wave_uniform(c);
    if (t_lo >= c) continue;
    if (s.gmask && !((s.gmask[start] >> fp) & 1ull)) continue;  // no row of the block touches the slot
    const unsigned long long own_bits = owner_bits(m, fp);  // (no loads inside the step loops)
    const bool chainy = !s.cmask || ((s.cmask[start] >> fp) & 1ull) != 0ull;  // multi-valued partner field
    const int i = wave_uniform(ud.x);
    float *rec = lat_row(m, i, fa) + sc * k;  // the slot's n row; z and w rows follow at RL, 2 RL
    int kk[G];
    bool act[G];
    float nc[G], zc[G], w[G];  // carries: valid in lane 0 of every row
    float sqc[G];              // sqrt(nc) while sq_valid (chain_sigma_w_forward)
    bool sq_valid = false;
#pragma unroll
    for (int g = 0; g < G; g++) {
      sqc[g] = 0.0f;
      const int grp = pass * G + g;
      act[g] = grp < groups;
      kk[g] = (act[g] ? grp : 0) * 4 + el;
      nc[g] = rec[LAT_N * RL + kk[g]];
      zc[g] = rec[LAT_Z * RL + kk[g]];
      w[g] = rec[LAT_W * RL + kk[g]];
    }
    const int4 *acol = s.haux + static_cast<int64_t>(start) * F + fp;  // + t*F
    const float2 *mcol = s.hmeta + start;                              // + t
    const int steps = (c - t_lo + kChainT - 1) / kChainT;

    if (!chainy) {
      // ---- three-stage pipeline over chunks of CH steps ----
      const int n_chunks = (steps + CH - 1) / CH;
      int4 axA[CH], axB[CH];
      float2 mtA[CH], mtB[CH];
      ChainTouch fB[CH], fC[CH];
      float vpB[CH][G], vpC[CH][G];
      auto load_raw = [&](int chunk, int4 (&ax)[CH], float2 (&mt)[CH]) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
          const int t = min(t_lo + (chunk * CH + j) * kChainT + tl, c - 1);  // past the end: repeats, unused
          ax[j] = acol[static_cast<int64_t>(t) * F];
          mt[j] = mcol[t];
        }
      };
      auto request_weights = [&](const int4 (&ax)[CH], const float2 (&mt)[CH], float (&vp)[CH][G],
                                 ChainTouch (&f)[CH]) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
          const int64_t off = haux_offset(ax[j].z, ax[j].w);
#pragma unroll
          for (int g = 0; g < G; g++) vp[j][g] = m.lat[off + kk[g]];
          f[j] = ChainTouch{mt[j].x, mt[j].y, __int_as_float(ax[j].x), ax[j].y};
        }
      };
      load_raw(0, axB, mtB);
      load_raw(1, axA, mtA);
      request_weights(axB, mtB, vpC, fC);
#pragma unroll
      for (int j = 0; j < CH; j++) { axB[j] = axA[j]; mtB[j] = mtA[j]; }
      for (int ch = 0; ch < n_chunks; ch++) {
        load_raw(ch + 2, axA, mtA);                // raw facts of chunk ch+2
        request_weights(axB, mtB, vpB, fB);        // partner weights of chunk ch+1
#pragma unroll
        for (int j = 0; j < CH; j++) {
          const int st = ch * CH + j;
          if (st < steps)
            chain_step<G>(m, own_bits, t_lo + st * kChainT + tl < c, l0, fC[j], vpC[j], act, w, nc, zc, sqc, sq_valid);
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
          fC[j] = fB[j];
          axB[j] = axA[j];
          mtB[j] = mtA[j];
#pragma unroll
          for (int g = 0; g < G; g++) vpC[j][g] = vpB[j][g];
        }
      }
    } else {
      // ---- a multi-valued partner field somewhere: facts two steps ahead, weights one ----
      int4 ax = acol[static_cast<int64_t>(min(t_lo + tl, c - 1)) * F];
      float2 mt = mcol[min(t_lo + tl, c - 1)];
      int4 axN = acol[static_cast<int64_t>(min(t_lo + kChainT + tl, c - 1)) * F];
      float2 mtN = mcol[min(t_lo + kChainT + tl, c - 1)];
      float vp[G];
#pragma unroll
      for (int g = 0; g < G; g++) vp[g] = m.lat[haux_offset(ax.z, ax.w) + kk[g]];
      for (int st = 0; st < steps; st++) {
        const int t = t_lo + st * kChainT + tl;
        float vpN[G];
#pragma unroll
        for (int g = 0; g < G; g++) vpN[g] = m.lat[haux_offset(axN.z, axN.w) + kk[g]];  // step st+1
        const int tNN = min(t_lo + (st + 2) * kChainT + tl, c - 1);                       // step st+2
        const int4 axNN = acol[static_cast<int64_t>(tNN) * F];
        const float2 mtNN = mcol[tNN];
        const int fl = ax.y;
        const bool live = t < c && owns_bit(own_bits, fl >> 8);
        if (!__any(live & ((fl & HF_CHAIN) != 0))) {
          chain_step<G>(m, own_bits, t < c, l0, ChainTouch{mt.x, mt.y, __int_as_float(ax.x), fl}, vp, act, w, nc, zc,
                        sqc, sq_valid);
        } else {
          sq_valid = false;
          // its 16 touches one after another, every lane of the row applying them to its own copy
          // of the running (n, z)
#pragma unroll
          for (int g = 0; g < G; g++) { nc[g] = row_first(nc[g]); zc[g] = row_first(zc[g]); }
          for (int tt = 0; tt < kChainT; tt++) {
            const int src = (lane & ~(kChainT - 1)) | tt;
            const int flt = __shfl(fl, src, 64);
            const float xot = __shfl(__int_as_float(ax.x), src, 64);
            const float tgt = __shfl(mt.x, src, 64), xmt = __shfl(mt.y, src, 64);
            float vpt[G];
#pragma unroll
            for (int g = 0; g < G; g++) vpt[g] = __shfl(vp[g], src, 64);
            const int fm = flt >> 8;
            if (t_lo + st * kChainT + tt >= c || !owns_bit(own_bits, fm)) continue;
            if (flt & HF_SIMPLE) {
#pragma unroll
              for (int g = 0; g < G; g++)
                ffm_touch(m.h, flt & HF_FIRST, tgt, xmt, xot, vpt[g], w[g], nc[g], zc[g]);
            } else if (flt & HF_CHAIN) {
              const int pt = s.occ2[start + t_lo + st * kChainT + tt].x;  // the touch's own entry
              const int r = s.row_of[pt];
              for (int qq = s.head[static_cast<int64_t>(r) * F + fp]; qq >= 0; qq = s.next[qq]) {
                if (qq == pt) continue;
                const float xq = rows.val[qq];
#pragma unroll
                for (int g = 0; g < G; g++) {
                  const float vq = m.lat[w_slot_offset(m, rows.feat[qq], fp, fm) + kk[g]];
                  ffm_touch(m.h, pt < qq, tgt, xmt, xq, vq, w[g], nc[g], zc[g]);
                }
              }
            }
          }
        }
        ax = axN; mt = mtN;
        axN = axNN; mtN = mtNN;
#pragma unroll
        for (int g = 0; g < G; g++) vp[g] = vpN[g];
      }
    }
    if (l0) {
#pragma unroll
      for (int g = 0; g < G; g++)
        if (act[g]) {
          rec[LAT_N * RL + kk[g]] = nc[g];
          rec[LAT_Z * RL + kk[g]] = zc[g];
        }
    }
  }
}

// One launch for the giant list (ModelDev::giant_min occurrences or more).  The first giant_blocks
// workgroups take the features with kGiantMin occurrences or more ONE chain per wave: a wave issues
// at most one VALU instruction every four cycles, so a wave carrying four interleaved chains needs
// ~2.4 us per 16-touch step -- for the 8600-touch chains of an 8-GPU job's 65536-row blocks that
// alone was 1.3 ms, the span of the whole update phase; one chain per wave is four times shorter,
// and there are few enough such features for the extra s_nop slots not to matter.  Dispatched first,
// the long chains also start first.  The other workgroups take the rest, G chains per wave.
// ph of `phases`: the touches that come from the rows of one row phase (engine_types.h).
template <int G>
__device__ __forceinline__ void ffm_chain_body(const ModelDev &m, const Rows &rows, const Scratch &s,
                                               int giant_blocks, int ph, int phases, unsigned bidx,
                                               unsigned gdim) {
  const unsigned w = wave_uniform(threadIdx.x >> 6);
  const int n_list = s.counters[CNT_NGIANT];
  if (static_cast<int>(bidx) < giant_blocks) {
    ffm_chain_items<1>(m, rows, s, s.giant, n_list, bidx * kUpdWaves + w, giant_blocks * kUpdWaves, ph, phases,
                       kGiantMin);
  } else
    ffm_chain_items<G>(m, rows, s, s.giant, n_list, (bidx - giant_blocks) * kUpdWaves + w,
                       (gdim - giant_blocks) * kUpdWaves, ph, phases, 0, giant_blocks > 0 ? kGiantMin : 0x7fffffff);
}
template <int G>
__global__ __launch_bounds__(kUpdThreads) void ffm_update_chain_kernel(ModelDev m, Rows rows,
                                                                       Scratch s, int giant_blocks,
                                                                       int ph, int phases) {
  ffm_chain_body<G>(m, rows, s, giant_blocks, ph, phases, blockIdx.x, gridDim.x);
}

// ---- FM -------------------------------------------------------------------------------------
// The same shape for FM::update_vector_nz (fm.cpp:80-101): work item = (very hot feature, pass over
// 16 of its factors); the per-touch inputs are the row's value, tmp_grad and factor sum (s.svx).
// (block of n_blocks: the workgroups of a launch that walk the very hot list)
template <int G>
__device__ __forceinline__ void fm_chain_body(const ModelDev &m, const Rows &rows, const Scratch &s,
                                              unsigned block, unsigned n_blocks) {
  const int k = m.n_factors;
  const int lane = threadIdx.x & 63;
  const int tl = lane & (kChainT - 1), el = lane >> 4;
  const bool l0 = tl == 0;
  const int groups = (k + 3) >> 2;
  const unsigned passes = (groups + G - 1) / G;
  const unsigned wave = block * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const unsigned n_waves = n_blocks * kUpdWaves;
  const unsigned n_huge = static_cast<unsigned>(s.counters[CNT_NHUGE]);
  const unsigned n_items = (n_huge + static_cast<unsigned>(s.counters[CNT_NGIANT])) * passes;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned li = item / passes;
    const int pass = static_cast<int>(item - li * passes);
    const int u = wave_uniform(li < n_huge ? s.huge[li] : s.giant[li - n_huge]);
    const int4 ud = s.udesc[u];
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    float *rec = lat_row(m, i, 0);
    int kk[G];
    bool act[G];
    float nc[G], zc[G], w[G], sqc[G];
    bool sq_valid = false;
#pragma unroll
    for (int g = 0; g < G; g++) {
      sqc[g] = 0.0f;
      const int e = (pass * G + g) * 4 + el;
      act[g] = e < k;
      kk[g] = act[g] ? e : 0;
      nc[g] = rec[LAT_N * k + kk[g]];
      zc[g] = rec[LAT_Z * k + kk[g]];
      w[g] = rec[LAT_W * k + kk[g]];
    }
    const int2 *ocol = s.occ2 + start;
    const int steps = (c + kChainT - 1) / kChainT;
    // pipeline: {entry, row} two steps ahead; value, tmp_grad and the row's factor sums one ahead
    int2 pr = ocol[min(tl, c - 1)];
    int2 prN = ocol[min(kChainT + tl, c - 1)];
    float x = rows.val[pr.x], tg = s.tg[pr.y];
    float sv[G];
#pragma unroll
    for (int g = 0; g < G; g++) sv[g] = s.svx[static_cast<int64_t>(pr.y) * k + kk[g]];
    for (int st = 0; st < steps; st++) {
      const int t = st * kChainT + tl;
      const float xN = rows.val[prN.x], tgN = s.tg[prN.y];
      float svN[G];
#pragma unroll
      for (int g = 0; g < G; g++) svN[g] = s.svx[static_cast<int64_t>(prN.y) * k + kk[g]];
      const int2 prNN = ocol[min((st + 2) * kChainT + tl, c - 1)];
      const bool live = t < c;
      float q[G], S[G], ga[G], arg0[G], nb[G], mm[G], mc[G], Z[G], nc_in[G];
      bool simple[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        nc_in[g] = nc[g];
        simple[g] = live && act[g];
        const float gr = tg * (x * sv[g] - w[g] * x * x);  // fm.cpp:84-95
        const float gg = gr * gr;
        ga[g] = simple[g] ? gr : -0.0f;
        q[g] = simple[g] ? gg : -0.0f;
        arg0[g] = gg;
        S[g] = nc[g] + q[g];
      }
      row_chain_add<G>(S, q);
#pragma unroll
      for (int g = 0; g < G; g++) nc[g] = row_mirror(S[g]);
      // every FM touch has its own g*g under the root: one square root per touch
      if (!chain_sigma_w_forward<G>(m.h, S, nc_in, simple, w, sqc, sq_valid, mm)) {
#pragma unroll
        for (int g = 0; g < G; g++) {
          nb[g] = row_left(nc_in[g], S[g]);
          arg0[g] = nb[g] + arg0[g];
        }
        chain_sigma_w<G>(m.h, nb, arg0, simple, w, mm);
        sq_valid = false;
      } else {
        sq_valid = true;
      }
#pragma unroll
      for (int g = 0; g < G; g++) {
        const float ms = simple[g] ? mm[g] : 0.0f;
        mc[g] = l0 ? 0.0f : ms;
        Z[g] = (zc[g] + ga[g]) - ms;
      }
      row_chain_addsub<G>(Z, ga, mc);
#pragma unroll
      for (int g = 0; g < G; g++) { zc[g] = row_mirror(Z[g]); sv[g] = svN[g]; }
      pr = prN; x = xN; tg = tgN;
      prN = prNN;
    }
    if (l0) {
#pragma unroll
      for (int g = 0; g < G; g++)
        if (act[g]) {
          rec[LAT_N * k + kk[g]] = nc[g];
          rec[LAT_Z * k + kk[g]] = zc[g];
        }
    }
  }
}
template <int G>
__global__ __launch_bounds__(kUpdThreads) void fm_update_chain_kernel(ModelDev m, Rows rows,
                                                                      Scratch s) {
  fm_chain_body<G>(m, rows, s, blockIdx.x, gridDim.x);
}

// The whole FM update of a block in ONE launch (no fork / join between streams: at FM's 0.3 ms per
// block the two event hops were 8 % of the step): workgroups [0, side_blocks) carry the bias chain
// and the linear update, the next chain_blocks the very hot features' chains (dispatched first: the
// longest), the rest the features in 2 .. huge_min rows.
__global__ __launch_bounds__(kUpdThreads) void fm_update_all_kernel(ModelDev m, Rows rows, Scratch s,
                                                                    int skip_once, int side_blocks,
                                                                    int chain_blocks) {
  const int b = blockIdx.x;
  if (b < side_blocks) {
    if (b == 0) {
      __builtin_amdgcn_s_setprio(3);  // one wave, n_rows dependent touches
      bias_update_body(m, 0, rows.n_rows, s);
    } else {
      linear_update_body(m, rows, s, b - 1, side_blocks - 1, 0, 1, skip_once);
    }
  } else if (b < side_blocks + chain_blocks) {
    fm_chain_body<4>(m, rows, s, b - side_blocks, chain_blocks);
  } else {
    fm_update_body(m, rows, s, 1, skip_once, b - side_blocks - chain_blocks, gridDim.x - side_blocks - chain_blocks);
  }
}

}  // namespace ftrl_dev
