// kernels_fold.h -- the block update of ONE (n, z) accumulator as reductions.
//
// With w and tmp_grad frozen over a block, the reference's per-sample step
//   sigma = (sqrtf(n + q) - sqrtf(n)) / alpha ;  z' = z + g - sigma*w ;  n' = n + g*g
// (src/model/ftrl_model.cpp:69-74, :81-84; src/model/ffm.cpp:112-120; src/model/fm.cpp:90-94)
// applied by every touching row in turn is, in exact arithmetic,
//   n_T = n_0 + sum g*g ,   z_T = z_0 + sum g - w * (sum of root differences) / alpha
// and for "plain" touches (q = g*g: linear, bias, FM, the first slot of an FFM pair) the root
// differences telescope to sqrtf(n_T) - sqrtf(n_0).  A touch of the
// ffm.cpp:118 kind (q = g2*g1) does not telescope: its difference sqrtf(n_t + q_t) - sqrtf(n_t) is
// taken on its own against the prefix sum n_t.  The floating-point TREE these kernels and the
// checker under oracle/ ("block update by reductions") share:
//   * the occurrences of a feature in the block, in row order, are cut into segments of kSeg
//     (rows of the block for the bias); inside a segment the sums run left to right from -0.0f
//     (the identity of fp addition), the segment totals are joined left to right:
//       B_0 = n_0,  B_{s+1} = B_s + P_s,  n_T = B_S ;  n_t = B_s + (partial P before touch t)
//   * from the accumulator's first :118 touch t0 on, every touch adds its root difference to D; the
//     plain touches before it telescope to sqrtf(n_t0) - sqrtf(n_0) (t0 = T when there is none);
//   * latent z_T = (z_0 + G) - ((head + D) / alpha) * w,  linear / bias z_T = z_0 + (G - sigma*w):
//     ONE alpha divide per accumulator and block.
// One touch is the reference's expression literally.  Accumulators that one row touches twice
// ("serial": s.cmask / UF_DUP) are not folded -- they keep the row-order walk (ffm_generic_body).
// No float atomics: every sum has ONE owner and a fixed order.
#pragma once
#include "engine_types.h"
#include "kernels_touch.h"

namespace ftrl_dev {

#ifndef FFM_FOLD_BRANCHY
#define FFM_FOLD_BRANCHY 0
#endif

// A sqrt operand of the per-touch root differences.  Strict: inside [2^-70, 2^96], where sqrt_fast
// is exact (two instructions to test).  Zero-tolerant: or +0 (a fresh model's n), for sqrt_fast0.
__device__ __forceinline__ bool fold_strict_ok(float x) {
  return __builtin_amdgcn_fmed3f(x, 0x1p-70f, 0x1p96f) == x;
}
__device__ __forceinline__ bool fold_zero_ok(float x) {
  return (__float_as_uint(x) == 0u) | (__builtin_amdgcn_fmed3f(x, 0x1p-70f, 0x1p96f) == x);
}

// d[i] = sqrtf(arg[i]) - sqrtf(nb[i]) for the lanes / entries that need it (the others get
// anything), correctly rounded roots: the five-instruction form behind ONE vote when every needed
// operand is comfortably normal, the same with +0 allowed behind a second vote, else sqrtf.
template <int N>
__device__ __forceinline__ void fold_root_diffs(const float (&arg)[N], const float (&nb)[N], const bool (&need)[N],
                                                float (&d)[N]) {
  bool ok = true;
#pragma unroll
  // (bitwise, not short-circuit: compiled as nested exec-mask branches these tests were a dozen
  // branches per group of four touches in the hot loop)
#if FFM_FOLD_BRANCHY
  for (int j = 0; j < N; j++) ok = ok && (!need[j] || (fold_strict_ok(arg[j]) && fold_strict_ok(nb[j])));
#else
  for (int j = 0; j < N; j++) ok = ok & (!need[j] | (static_cast<int>(fold_strict_ok(arg[j])) & static_cast<int>(fold_strict_ok(nb[j]))));
#endif
  if (__all(ok)) {
#pragma unroll
    for (int j = 0; j < N; j++) d[j] = sqrt_fast(arg[j]) - sqrt_fast(nb[j]);
    return;
  }
  ok = true;
#pragma unroll
#if FFM_FOLD_BRANCHY
  for (int j = 0; j < N; j++) ok = ok && (!need[j] || (fold_zero_ok(arg[j]) && fold_zero_ok(nb[j])));
#else
  for (int j = 0; j < N; j++) ok = ok & (!need[j] | (static_cast<int>(fold_zero_ok(arg[j])) & static_cast<int>(fold_zero_ok(nb[j]))));
#endif
  if (__all(ok)) {
#pragma unroll
    for (int j = 0; j < N; j++) d[j] = sqrt_fast0(arg[j]) - sqrt_fast0(nb[j]);
  } else {
#pragma unroll
    for (int j = 0; j < N; j++) d[j] = sqrtf(arg[j]) - sqrtf(nb[j]);
  }
}

// One accumulator per lane.
struct Fold {
  float P, G, D;        // the running segment: sum g*g, sum g, sum of root differences
  float B, Gacc, Dacc;  // the segments before it (B starts at n_0)
  float ncap;           // n_t at the first :118 touch
  bool any, seen, head_plain;
  __device__ __forceinline__ void init(float n0) {
    P = G = D = Gacc = Dacc = -0.0f;
    B = n0;
    ncap = 0.0f;
    any = seen = head_plain = false;
  }
  // a segment ends (joining an empty one adds -0.0f: nothing)
  __device__ __forceinline__ void flush() {
    B = B + P;
    Gacc = Gacc + G;
    Dacc = Dacc + D;
    P = G = D = -0.0f;
  }
  // plain touches only (linear, bias, FM): gradient g of a live touch
  __device__ __forceinline__ void plain(bool live, float g) {
    const float gg = g * g;
    G = live ? G + g : G;
    P = live ? P + gg : P;
    any = any || live;
    head_plain = any;
  }
};

// N consecutive touches of one FFM accumulator per lane (lane = element).  live: the touch exists
// for this lane's slot; first: live and its own entry is the pair's first (ffm.cpp:112-115; a live
// touch that is not first is the pair's second: :117-120 with the :118 quirk); g = tmp_grad * vp * x
// as the reference associates it.
template <int N>
__device__ __forceinline__ void fold_ffm_group(Fold &a, float w, const bool (&live)[N],
                                               const bool (&first)[N], const float (&tg)[N],
                                               const float (&x)[N], const float (&vp)[N]) {
  float g[N], gg[N];
  bool quirk[N], anyq = false;
#pragma unroll
  for (int j = 0; j < N; j++) {
    g[j] = tg[j] * vp[j] * x[j];
    gg[j] = g[j] * g[j];
    quirk[j] = live[j] & !first[j];
    anyq = anyq | quirk[j];
  }
  if (__any(a.seen | anyq)) {
    float arg[N], nb[N], d[N];
    bool need[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const float nt = a.B + a.P;
      const bool flip = quirk[j] & !a.seen;
      if (__any(flip)) a.ncap = flip ? nt : a.ncap;
      a.seen = a.seen | flip;
      need[j] = a.seen & live[j];
      const float g1 = tg[j] * w * x[j];  // the pair's first entry's gradient (ffm.cpp:112)
      arg[j] = nt + (first[j] ? gg[j] : g[j] * g1);  // ffm.cpp:113 / :118
      nb[j] = nt;
      a.P = live[j] ? a.P + gg[j] : a.P;
    }
    fold_root_diffs<N>(arg, nb, need, d);
#pragma unroll
    for (int j = 0; j < N; j++) a.D = need[j] ? a.D + d[j] : a.D;
  } else {
#pragma unroll
    for (int j = 0; j < N; j++) a.P = live[j] ? a.P + gg[j] : a.P;
  }
#pragma unroll
  for (int j = 0; j < N; j++) {
    a.G = live[j] ? a.G + g[j] : a.G;
    if (!a.any) a.head_plain = first[j];  // (kept from the first live touch on)
    a.any = a.any | live[j];
  }
}

// The same N touches when the BLOCK is regular -- every row holds exactly one entry per field, in field
// order, ids inside their fields' ranges (group_keys_kernel: CNT_IRREGULAR stayed 0) -- so that a touch's
// flags are the lane's: live = the slot is not the feature's own field's (lv), and the own entry is the
// pair's first exactly when its field is the smaller one (q118 = lv and own field > partner field, the
// learning variant aside).  Same arithmetic in the same order for every live lane; no per-touch flag is
// read, decoded or selected on.  (Lanes that are not live compute on whatever the tile holds and are
// never stored.)
template <int N>
__device__ __forceinline__ void fold_ffm_group_regular(Fold &a, float w, bool lv, bool q118, const float (&tg)[N],
                                                       const float (&x)[N], const float (&vp)[N]) {
  float g[N], gg[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    g[j] = tg[j] * vp[j] * x[j];
    gg[j] = g[j] * g[j];
  }
  if (__any(q118)) {
    float arg[N], nb[N], d[N];
    bool need[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const float nt = a.B + a.P;
      a.ncap = (q118 & !a.seen) ? nt : a.ncap;  // (the lane's first touch)
      a.seen = a.seen | q118;
      need[j] = q118;
      const float g1 = tg[j] * w * x[j];  // the pair's first entry's gradient (ffm.cpp:112)
      arg[j] = nt + g[j] * g1;            // ffm.cpp:118 (only the q118 lanes' values are used)
      nb[j] = nt;
      a.P = a.P + gg[j];
    }
    fold_root_diffs<N>(arg, nb, need, d);
#pragma unroll
    for (int j = 0; j < N; j++) a.D = q118 ? a.D + d[j] : a.D;
  } else {
#pragma unroll
    for (int j = 0; j < N; j++) a.P = a.P + gg[j];
  }
#pragma unroll
  for (int j = 0; j < N; j++) a.G = a.G + g[j];
  a.head_plain = a.any ? a.head_plain : !q118;
  a.any = a.any | lv;
}

// ONE touch of E accumulators held by one lane that share the touch's flags (the factors of one
// slot), for features with at most kSeg occurrences: a single segment, so B stays n_0 and the
// running sums are the totals.  any / seen / head_plain are the lane's (the E factors move together).
template <int E>
struct FoldFew {
  float P[E], G[E], D[E], ncap[E];
  bool any, seen, head_plain;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < E; i++) { P[i] = G[i] = D[i] = -0.0f; ncap[i] = 0.0f; }
    any = seen = head_plain = false;
  }
  // first: the own entry is the pair's first (or the learning variant: g2*g2 at ffm.cpp:118)
  __device__ __forceinline__ void touch(const float (&n0)[E], const float (&w)[E], bool live, bool first,
                                        float tg, float x, const float (&vp)[E]) {
    float g[E], gg[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
      g[i] = tg * vp[i] * x;
      gg[i] = g[i] * g[i];
    }
    const bool quirk = live && !first;
    if (__any(seen || quirk)) {
      const bool flip = quirk && !seen;
      float arg[E], nb[E], d[E];
      bool need[E];
      seen = seen || flip;
#pragma unroll
      for (int i = 0; i < E; i++) {
        const float nt = n0[i] + P[i];
        ncap[i] = flip ? nt : ncap[i];
        const float g1 = tg * w[i] * x;  // the pair's first entry's gradient (ffm.cpp:112)
        arg[i] = nt + (first ? gg[i] : g[i] * g1);  // ffm.cpp:113 / :118
        nb[i] = nt;
        need[i] = seen && live;
      }
      fold_root_diffs<E>(arg, nb, need, d);
#pragma unroll
      for (int i = 0; i < E; i++) D[i] = need[i] ? D[i] + d[i] : D[i];
    }
#pragma unroll
    for (int i = 0; i < E; i++) {
      P[i] = live ? P[i] + gg[i] : P[i];
      G[i] = live ? G[i] + g[i] : G[i];
    }
    if (live && !any) { any = true; head_plain = first; }
  }
  // The same touch in a regular block (fold_ffm_group_regular): the lane's slot is live (lv) for every
  // touch, and every touch is of the ffm.cpp:118 kind (q118) or none is.
  __device__ __forceinline__ void touch_regular(const float (&n0)[E], const float (&w)[E], bool lv, bool q118,
                                                float tg, float x, const float (&vp)[E]) {
    float g[E], gg[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
      g[i] = tg * vp[i] * x;
      gg[i] = g[i] * g[i];
    }
    if (__any(q118)) {
      float arg[E], nb[E], d[E];
      bool need[E];
#pragma unroll
      for (int i = 0; i < E; i++) {
        const float nt = n0[i] + P[i];
        ncap[i] = (q118 & !seen) ? nt : ncap[i];
        const float g1 = tg * w[i] * x;  // the pair's first entry's gradient (ffm.cpp:112)
        arg[i] = nt + g[i] * g1;         // ffm.cpp:118
        nb[i] = nt;
        need[i] = q118;
      }
      seen = seen | q118;
      fold_root_diffs<E>(arg, nb, need, d);
#pragma unroll
      for (int i = 0; i < E; i++) D[i] = q118 ? D[i] + d[i] : D[i];
    }
#pragma unroll
    for (int i = 0; i < E; i++) {
      P[i] = P[i] + gg[i];
      G[i] = G[i] + g[i];
    }
    head_plain = any ? head_plain : (lv & !q118);
    any = any | lv;
  }
  // (n, z) in, (n_T, z_T) out for the lane's E accumulators; false when no touch reached them
  __device__ __forceinline__ bool finish(const Hyper &h, const float (&w)[E], float (&n)[E], float (&z)[E]) {
    float nT[E], sa[E], sb[E], S[E], q[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
      nT[i] = n[i] + P[i];
      sa[i] = seen ? ncap[i] : nT[i];
      sb[i] = n[i];
    }
    float ra[E], rb[E];
    sqrt_cr_n<E>(sa, ra);
    sqrt_cr_n<E>(sb, rb);
#pragma unroll
    for (int i = 0; i < E; i++) {
      float t = -0.0f;
      t = head_plain ? t + (ra[i] - rb[i]) : t;
      S[i] = t + D[i];
    }
    div_alpha_n<E>(h, S, q);
#pragma unroll
    for (int i = 0; i < E; i++) {
      if (any) {
        z[i] = (z[i] + G[i]) - q[i] * w[i];
        n[i] = nT[i];
      }
    }
    return any;
  }
};

// The end of a latent accumulator's block: joins the last segment; (n, z) in, (n_T, z_T) out.
// Returns false (and leaves n, z alone) when no touch reached the accumulator.
__device__ __forceinline__ bool fold_finish_latent(const Hyper &h, Fold &a, float w, float &n, float &z) {
  a.flush();
  const float ncap = a.seen ? a.ncap : a.B;
  const float hd = sqrt_cr(ncap) - sqrt_cr(n);
  float S = -0.0f;
  S = a.head_plain ? S + hd : S;
  S = S + a.Dacc;
  const float m = div_alpha(h, S) * w;
  if (a.any) {
    z = (z + a.Gacc) - m;
    n = a.B;
  }
  return a.any;
}
// ... of a linear / bias accumulator (every touch plain): z += g - sigma*w summed is z + (G - sigma*w)
__device__ __forceinline__ bool fold_finish_linear(const Hyper &h, Fold &a, float w, float &n, float &z) {
  a.flush();
  const float si = div_alpha(h, sqrt_cr(a.B) - sqrt_cr(n));
  if (a.any) {
    z = z + (a.Gacc - si * w);
    n = a.B;
  }
  return a.any;
}

}  // namespace ftrl_dev
