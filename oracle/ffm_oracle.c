/*
 * ffm_oracle.c -- CPU restatement of the Ftrl-FFM hot path.  TEST INFRASTRUCTURE ONLY: see
 * ffm_oracle.h.  Parity PINNED against the compiled reference (oracle/_ref) and tests/golden/.
 *
 * Build: gcc -O3 -ffp-contract=off (the reference is built -O3 for baseline x86-64, i.e. without
 * FMA: /root/reference/CMakeLists.txt:8-11), so every a*b+c below rounds twice, as there.
 */
#include "ffm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct fo_model {
  int model_type, n_feats, n_fields, n_factors;
  int64_t row_len;
  float w_alpha, w_beta, w_l1, w_l2;
  float bias3[3]; /* bias, bias_n, bias_z  (ftrl_model.h:36,45-46) */
  float *lin_w, *lin_n, *lin_z; /* ftrl_model.h:37,47-48 */
  float *vec_w, *vec_n, *vec_z; /* ffm.h:25,30-31 / fm.h:20,25-26 */
  float *sum_vx;                /* fm.h:24 */
  pthread_mutex_t *locks;       /* ftrl_model.h:49, ffm.h:32 (threaded baseline only) */
  pthread_mutex_t bias_lock;    /* ftrl_model.h:50 */
  int learn;                    /* fo_set_variant: SURVEY.md 8(f) rank 4, NOT reference behaviour */
};

fo_model *fo_create(int model_type, int n_feats, int n_fields, int n_factors, float w_alpha,
                    float w_beta, float w_l1, float w_l2) {
  fo_model *m = (fo_model *)calloc(1, sizeof(fo_model));
  if (!m) return NULL;
  m->model_type = model_type;
  m->n_feats = n_feats;
  m->n_fields = n_fields;
  m->n_factors = n_factors;
  m->w_alpha = w_alpha;
  m->w_beta = w_beta;
  m->w_l1 = w_l1;
  m->w_l2 = w_l2;
  m->row_len = model_type == FO_FFM ? (int64_t)n_fields * n_factors
               : model_type == FO_FM ? n_factors
                                     : 0;
  size_t nf = (size_t)n_feats;
  m->lin_w = (float *)calloc(nf ? nf : 1, sizeof(float));
  m->lin_n = (float *)calloc(nf ? nf : 1, sizeof(float));
  m->lin_z = (float *)calloc(nf ? nf : 1, sizeof(float));
  size_t nv = nf * (size_t)m->row_len;
  m->vec_w = (float *)calloc(nv ? nv : 1, sizeof(float));
  m->vec_n = (float *)calloc(nv ? nv : 1, sizeof(float));
  m->vec_z = (float *)calloc(nv ? nv : 1, sizeof(float));
  m->sum_vx = (float *)calloc((size_t)(n_factors > 0 ? n_factors : 1), sizeof(float));
  pthread_mutex_init(&m->bias_lock, NULL);
  return m;
}

void fo_destroy(fo_model *m) {
  if (!m) return;
  free(m->lin_w); free(m->lin_n); free(m->lin_z);
  free(m->vec_w); free(m->vec_n); free(m->vec_z);
  free(m->sum_vx);
  if (m->locks) {
    for (int i = 0; i < m->n_feats; i++) pthread_mutex_destroy(&m->locks[i]);
    free(m->locks);
  }
  pthread_mutex_destroy(&m->bias_lock);
  free(m);
}

/* Opt-in variant that lets the latent factors train (off = the reference, bit for bit):
 * (1) the lazy refresh keeps a slot's initial weight until its first gradient (n > 0), instead of
 *     overwriting it with W(0, 0) = 0 (ffm.cpp:72-88, fm.cpp:69-78);
 * (2) the second slot's step size uses g2*g2, not g2*g1 (ffm.cpp:118). */
void fo_set_variant(fo_model *m, int learn) { m->learn = learn != 0; }

/* latent refresh under the variant rule */
static inline float mzw_latent(const fo_model *m, float n, float z, float w_old);

float *fo_bias3(fo_model *m) { return m->bias3; }
float *fo_lin_w(fo_model *m) { return m->lin_w; }
float *fo_lin_n(fo_model *m) { return m->lin_n; }
float *fo_lin_z(fo_model *m) { return m->lin_z; }
float *fo_vec_w(fo_model *m) { return m->vec_w; }
float *fo_vec_n(fo_model *m) { return m->vec_n; }
float *fo_vec_z(fo_model *m) { return m->vec_z; }
int64_t fo_row_len(const fo_model *m) { return m->row_len; }

/* utils.h:15-18: x > 0 ? 1 : -1  (so sgn(0) = -1, pinned by tests/test_utils.cpp:13-18) */
float fo_sgn(float x) { return x > 0 ? 1.0f : -1.0f; }

/* utils.h:20-23 with T = float: 1 / (1 + std::exp(-x)), std::exp(float) is expf */
float fo_sigmoid(float x) { return 1 / (1 + expf(-x)); }

/* eval/loss.h:8-12: sigmoid in double, -y*log(s) - (1-y)*log(1-s) */
double fo_loss(int y, double logit) {
  const double s = 1 / (1 + exp(-logit));
  return -y * log(s) - (1 - y) * log(1 - s);
}

/* ftrl_model.h:28-33, T = float.  The 0.0 / -1.0 literals promote numerator and denominator to
 * double for the divide; the result narrows to float on return. */
static inline float mzw(const fo_model *m, float n, float z) {
  if (fabsf(z) <= m->w_l1) return (float)0.0;
  const float num = z - fo_sgn(z) * m->w_l1;
  const float den = m->w_l2 + (m->w_beta + sqrtf(n)) / m->w_alpha;
  return (float)(-1.0 * (double)num / (double)den);
}
float fo_maybe_zero_weight(const fo_model *m, float n, float z) { return mzw(m, n, z); }

static inline float mzw_latent(const fo_model *m, float n, float z, float w_old) {
  if (m->learn && !(n > 0.0f)) return w_old;
  return mzw(m, n, z);
}

/* ftrl_model.cpp:36-42 (feat only) and ffm.cpp:30-36 (field too) */
static inline int in_range(const fo_model *m, int field, int feat) {
  if (feat < 0 || feat >= m->n_feats) return 0;
  if (m->model_type == FO_FFM && (field < 0 || field >= m->n_fields)) return 0;
  return 1;
}

/* Filtered row view: indices of surviving entries, in row order. */
typedef struct {
  int n;
  const int32_t *field, *feat;
  const float *val;
  int *idx;
} rowview;

static int rv_build(const fo_model *m, rowview *rv, int nnz, const int32_t *field,
                    const int32_t *feat, const float *val, int *idx_buf) {
  rv->field = field; rv->feat = feat; rv->val = val; rv->idx = idx_buf; rv->n = 0;
  for (int p = 0; p < nnz; p++)
    if (in_range(m, field[p], feat[p])) idx_buf[rv->n++] = p;
  return rv->n;
}
#define RV_FIELD(rv, a) ((rv)->field[(rv)->idx[a]])
#define RV_FEAT(rv, a) ((rv)->feat[(rv)->idx[a]])
#define RV_VAL(rv, a) ((rv)->val[(rv)->idx[a]])

/* ftrl_model.cpp:52-59 */
static void update_linear_w(fo_model *m, const rowview *rv) {
  for (int a = 0; a < rv->n; a++) {
    const int i = RV_FEAT(rv, a);
    m->lin_w[i] = mzw(m, m->lin_n[i], m->lin_z[i]);
  }
}
/* ftrl_model.cpp:61-64 */
static void update_bias(fo_model *m) { m->bias3[0] = mzw(m, m->bias3[1], m->bias3[2]); }

/* ftrl_model.cpp:44-50: std::accumulate from bias, acc + lin_w[i]*x, row order */
static float compute_linear_logit(const fo_model *m, const rowview *rv) {
  float acc = m->bias3[0];
  for (int a = 0; a < rv->n; a++) acc = acc + m->lin_w[RV_FEAT(rv, a)] * RV_VAL(rv, a);
  return acc;
}

/* One FTRL accumulator step, ftrl_model.cpp:69-74 / :81-84:
 *   g = tmp_grad*x ; s = (sqrtf(n+g*g)-sqrtf(n))/alpha ; z += g - s*w ; n += g*g */
static inline void nz_step(const fo_model *m, float w, float g, float *n, float *z) {
  const float ni = *n;
  const float si = (sqrtf(ni + g * g) - sqrtf(ni)) / m->w_alpha;
  *z += g - si * w;
  *n += g * g;
}

/* ftrl_model.cpp:66-77 */
static void update_linear_nz(fo_model *m, const rowview *rv, float tmp_grad) {
  for (int a = 0; a < rv->n; a++) {
    const int i = RV_FEAT(rv, a);
    nz_step(m, m->lin_w[i], tmp_grad * RV_VAL(rv, a), &m->lin_n[i], &m->lin_z[i]);
  }
}
/* ftrl_model.cpp:79-85 */
static void update_bias_nz(fo_model *m, float tmp_grad) {
  nz_step(m, m->bias3[0], tmp_grad, &m->bias3[1], &m->bias3[2]);
}

/* ---------------- FFM ---------------- */

/* ffm.cpp:72-88 */
static void ffm_update_vector_w(fo_model *m, const rowview *rv) {
  const int k = m->n_factors;
  const int64_t L = m->row_len;
  for (int a = 0; a < rv->n; a++)
    for (int b = a + 1; b < rv->n; b++) {
      const int field1 = RV_FIELD(rv, a), i = RV_FEAT(rv, a);
      const int field2 = RV_FIELD(rv, b), j = RV_FEAT(rv, b);
      for (int f = 0; f < k; f++) {
        const int64_t f1 = i * L + (int64_t)field2 * k + f;
        m->vec_w[f1] = mzw_latent(m, m->vec_n[f1], m->vec_z[f1], m->vec_w[f1]);
        const int64_t f2 = j * L + (int64_t)field1 * k + f;
        m->vec_w[f2] = mzw_latent(m, m->vec_n[f2], m->vec_z[f2], m->vec_w[f2]);
      }
    }
}

/* ffm.cpp:57-70: linear logit, then per pair inner_product(init 0.0f) * x1 * x2 */
static float compute_ffm_logit(const fo_model *m, const rowview *rv) {
  const int k = m->n_factors;
  const int64_t L = m->row_len;
  float result = compute_linear_logit(m, rv);
  for (int a = 0; a < rv->n; a++)
    for (int b = a + 1; b < rv->n; b++) {
      const int field1 = RV_FIELD(rv, a), i = RV_FEAT(rv, a);
      const int field2 = RV_FIELD(rv, b), j = RV_FEAT(rv, b);
      const float x1 = RV_VAL(rv, a), x2 = RV_VAL(rv, b);
      const float *vi = m->vec_w + i * L + (int64_t)field2 * k;
      const float *vj = m->vec_w + j * L + (int64_t)field1 * k;
      float dot = 0.0f;
      for (int f = 0; f < k; f++) dot = dot + vi[f] * vj[f];
      result += dot * x1 * x2;
    }
  return result;
}

/* The per-pair, per-factor body of ffm.cpp:102-121, including the :118 quirk
 * (sqrtf(n2 + g2*g1)).  Writes through immediately: single-threaded this equals the reference's
 * copy-back at :129-132 because a pair's 4k slots are distinct unless i == j (which deadlocks
 * the reference). */
static inline void ffm_pair_step(const fo_model *m, float tmp_grad, float x, float vif1,
                                 float vif2, float *n1, float *z1, float *n2, float *z2) {
  const float v_nif1 = *n1, v_zif1 = *z1, v_nif2 = *n2, v_zif2 = *z2;
  const float v_gif1 = tmp_grad * vif2 * x;
  const float v_sif1 = (sqrtf(v_nif1 + v_gif1 * v_gif1) - sqrtf(v_nif1)) / m->w_alpha;
  const float zi1 = v_zif1 + v_gif1 - v_sif1 * vif1;
  const float ni1 = v_nif1 + v_gif1 * v_gif1;
  const float v_gif2 = tmp_grad * vif1 * x;
  const float v_sif2 =
      (sqrtf(v_nif2 + (m->learn ? v_gif2 * v_gif2 : v_gif2 * v_gif1)) - sqrtf(v_nif2)) / m->w_alpha;
  const float zi2 = v_zif2 + v_gif2 - v_sif2 * vif2;
  const float ni2 = v_nif2 + v_gif2 * v_gif2;
  *z1 = zi1; *n1 = ni1; *z2 = zi2; *n2 = ni2;
}

/* ffm.cpp:90-136 */
static void ffm_update_vector_nz(fo_model *m, const rowview *rv, float tmp_grad) {
  const int k = m->n_factors;
  const int64_t L = m->row_len;
  float tn1[256], tz1[256], tn2[256], tz2[256];
  for (int a = 0; a < rv->n; a++)
    for (int b = a + 1; b < rv->n; b++) {
      const int field1 = RV_FIELD(rv, a), i = RV_FEAT(rv, a);
      const int field2 = RV_FIELD(rv, b), j = RV_FEAT(rv, b);
      const float x = RV_VAL(rv, a) * RV_VAL(rv, b);
      const int64_t o1 = i * L + (int64_t)field2 * k, o2 = j * L + (int64_t)field1 * k;
      if (o1 == o2) {
        /* The same (field, id) twice in one row: both sides of the pair are ONE slot.  The
         * reference never gets here -- std::scoped_lock on the same mutex twice (ffm.cpp:99-101,
         * :124) deadlocks, SURVEY.md section 0 item 3 -- so nothing pins this case; the block
         * algorithm's rule applies: every touch of a slot is applied in order to the running
         * (n, z): first the pair's i-side step (:112-115), then its j-side step (:117-120) on the
         * result, with w frozen. */
        for (int f = 0; f < k; f++) {
          const float w = m->vec_w[o1 + f];
          const float g = tmp_grad * w * x; /* g1 == g2: both partner weights are this slot's */
          float nn = m->vec_n[o1 + f], zz = m->vec_z[o1 + f];
          const float s1 = (sqrtf(nn + g * g) - sqrtf(nn)) / m->w_alpha;
          zz = zz + g - s1 * w;
          nn = nn + g * g;
          const float s2 = (sqrtf(nn + g * g) - sqrtf(nn)) / m->w_alpha;
          zz = zz + g - s2 * w;
          nn = nn + g * g;
          m->vec_n[o1 + f] = nn; m->vec_z[o1 + f] = zz;
        }
      } else if (k <= 256) {
        /* read phase into temporaries, then copy back (ffm.cpp:96-132) */
        for (int f = 0; f < k; f++) {
          tn1[f] = m->vec_n[o1 + f]; tz1[f] = m->vec_z[o1 + f];
          tn2[f] = m->vec_n[o2 + f]; tz2[f] = m->vec_z[o2 + f];
          ffm_pair_step(m, tmp_grad, x, m->vec_w[o1 + f], m->vec_w[o2 + f], &tn1[f], &tz1[f],
                        &tn2[f], &tz2[f]);
        }
        for (int f = 0; f < k; f++) { m->vec_z[o1 + f] = tz1[f]; m->vec_n[o1 + f] = tn1[f]; }
        for (int f = 0; f < k; f++) { m->vec_z[o2 + f] = tz2[f]; m->vec_n[o2 + f] = tn2[f]; }
      } else {
        for (int f = 0; f < k; f++)
          ffm_pair_step(m, tmp_grad, x, m->vec_w[o1 + f], m->vec_w[o2 + f], &m->vec_n[o1 + f],
                        &m->vec_z[o1 + f], &m->vec_n[o2 + f], &m->vec_z[o2 + f]);
      }
    }
}

/* ---------------- FM ---------------- */

/* fm.cpp:69-78 */
static void fm_update_vector_w(fo_model *m, const rowview *rv) {
  const int k = m->n_factors;
  for (int a = 0; a < rv->n; a++) {
    const int64_t o = (int64_t)RV_FEAT(rv, a) * k;
    for (int f = 0; f < k; f++)
      m->vec_w[o + f] = mzw_latent(m, m->vec_n[o + f], m->vec_z[o + f], m->vec_w[o + f]);
  }
}

/* fm.cpp:40-67; sum_vx (k floats) receives the per-factor sums when it is non-NULL */
static float compute_fm_logit(const fo_model *m, const rowview *rv, float *sum_vx) {
  const int k = m->n_factors;
  float result = compute_linear_logit(m, rv);
  for (int f = 0; f < k; f++) {
    float s_vx = 0.0;
    float sum_sqr = 0.0;
    for (int a = 0; a < rv->n; a++) {
      const float vx = m->vec_w[(int64_t)RV_FEAT(rv, a) * k + f] * RV_VAL(rv, a);
      s_vx += vx;
      sum_sqr += vx * vx;
    }
    if (sum_vx) sum_vx[f] = s_vx;
    result += 0.5f * (s_vx * s_vx - sum_sqr);
  }
  return result;
}

/* fm.cpp:84-95 for one (feature, x) occurrence */
static inline void fm_feat_step(fo_model *m, int i, float x, float tmp_grad, const float *sum_vx) {
  const int k = m->n_factors;
  const int64_t o = (int64_t)i * k;
  for (int f = 0; f < k; f++) {
    const float vif = m->vec_w[o + f];
    const float v_nif = m->vec_n[o + f];
    const float v_zif = m->vec_z[o + f];
    const float s_vx = sum_vx[f];
    const float v_gif = tmp_grad * (x * s_vx - vif * x * x);
    const float v_sif = (sqrtf(v_nif + v_gif * v_gif) - sqrtf(v_nif)) / m->w_alpha;
    m->vec_z[o + f] = v_zif + v_gif - v_sif * vif;
    m->vec_n[o + f] = v_nif + v_gif * v_gif;
  }
}

/* fm.cpp:80-101 */
static void fm_update_vector_nz(fo_model *m, const rowview *rv, float tmp_grad,
                                const float *sum_vx) {
  for (int a = 0; a < rv->n; a++) fm_feat_step(m, RV_FEAT(rv, a), RV_VAL(rv, a), tmp_grad, sum_vx);
}

/* ---------------- train / predict ---------------- */

#define FO_STACK_NNZ 4096

float fo_train(fo_model *m, int nnz, const int32_t *field, const int32_t *feat, const float *val,
               int label) {
  int stack_idx[FO_STACK_NNZ];
  int *idx = nnz <= FO_STACK_NNZ ? stack_idx : (int *)malloc(sizeof(int) * (size_t)nnz);
  rowview rv;
  rv_build(m, &rv, nnz, field, feat, val, idx); /* remove_out_range */
  update_linear_w(m, &rv);
  update_bias(m);
  float logit;
  if (m->model_type == FO_FFM) {
    ffm_update_vector_w(m, &rv);
    logit = compute_ffm_logit(m, &rv);
  } else if (m->model_type == FO_FM) {
    fm_update_vector_w(m, &rv);
    logit = compute_fm_logit(m, &rv, m->sum_vx);
  } else {
    logit = compute_linear_logit(m, &rv);
  }
  const float tmp_grad = fo_sigmoid(logit) - (float)label;
  update_linear_nz(m, &rv, tmp_grad);
  update_bias_nz(m, tmp_grad);
  if (m->model_type == FO_FFM) ffm_update_vector_nz(m, &rv, tmp_grad);
  else if (m->model_type == FO_FM) fm_update_vector_nz(m, &rv, tmp_grad, m->sum_vx);
  if (idx != stack_idx) free(idx);
  return logit;
}

float fo_predict(fo_model *m, int nnz, const int32_t *field, const int32_t *feat, const float *val,
                 int output_prob) {
  int stack_idx[FO_STACK_NNZ];
  int *idx = nnz <= FO_STACK_NNZ ? stack_idx : (int *)malloc(sizeof(int) * (size_t)nnz);
  rowview rv;
  rv_build(m, &rv, nnz, field, feat, val, idx);
  float logit;
  if (m->model_type == FO_FFM) logit = compute_ffm_logit(m, &rv);
  else if (m->model_type == FO_FM) logit = compute_fm_logit(m, &rv, NULL);
  else logit = compute_linear_logit(m, &rv);
  if (idx != stack_idx) free(idx);
  return output_prob ? fo_sigmoid(logit) : logit;
}

double fo_train_rows(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                     const int32_t *feat, const float *val, const int32_t *label,
                     float *logit_out) {
  double tmp_loss = 0.0;
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r], e = row_ptr[r + 1];
    const float logit = fo_train(m, e - b, field + b, feat + b, val + b, label[r]);
    if (logit_out) logit_out[r] = logit;
    tmp_loss += fo_loss(label[r], logit); /* ftrl_online.cpp:75-76 */
  }
  return tmp_loss;
}

double fo_predict_batch(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                        const int32_t *feat, const float *val, const int32_t *label,
                        int output_prob, float *out) {
  double tmp_loss = 0.0;
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r], e = row_ptr[r + 1];
    const float logit = fo_predict(m, e - b, field + b, feat + b, val + b, 0);
    if (out) out[r] = output_prob ? fo_sigmoid(logit) : logit;
    if (label) tmp_loss += fo_loss(label[r], logit); /* evaluate.cpp:28-29 */
  }
  return tmp_loss;
}

/* The ROW-WALK block update (rounds 1-4 of this repo, kept as a yardstick): three sweeps over the
 * batch; each sweep visits rows in order and, inside a row, follows the reference's statement
 * order, so every touch of an accumulator lands on the running (n, z).  In exact arithmetic it is
 * what fo_train_batch below computes with reductions; tests bound the rounding distance. */
double fo_train_batch_rowwalk(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                      const int32_t *feat, const float *val, const int32_t *label,
                      float *logit_out) {
  const int k = m->n_factors;
  int max_nnz = 1;
  for (int r = 0; r < n_rows; r++)
    if (row_ptr[r + 1] - row_ptr[r] > max_nnz) max_nnz = row_ptr[r + 1] - row_ptr[r];
  int *idx = (int *)malloc(sizeof(int) * (size_t)max_nnz);
  float *tg = (float *)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1));
  float *svx = NULL;
  if (m->model_type == FO_FM)
    svx = (float *)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * (size_t)(k > 0 ? k : 1));
  rowview rv;
  double tmp_loss = 0.0;
  /* sweep 1: lazy refresh of everything the batch touches, from the batch-start (n,z) */
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r];
    rv_build(m, &rv, row_ptr[r + 1] - b, field + b, feat + b, val + b, idx);
    update_linear_w(m, &rv);
    if (m->model_type == FO_FFM) ffm_update_vector_w(m, &rv);
    else if (m->model_type == FO_FM) fm_update_vector_w(m, &rv);
  }
  if (n_rows > 0) update_bias(m);
  /* sweep 2: forward with frozen weights */
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r];
    rv_build(m, &rv, row_ptr[r + 1] - b, field + b, feat + b, val + b, idx);
    float logit;
    if (m->model_type == FO_FFM) logit = compute_ffm_logit(m, &rv);
    else if (m->model_type == FO_FM) logit = compute_fm_logit(m, &rv, svx + (size_t)r * k);
    else logit = compute_linear_logit(m, &rv);
    tg[r] = fo_sigmoid(logit) - (float)label[r];
    if (logit_out) logit_out[r] = logit;
    tmp_loss += fo_loss(label[r], logit);
  }
  /* sweep 3: accumulator updates in row order, w and tmp_grad frozen */
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r];
    rv_build(m, &rv, row_ptr[r + 1] - b, field + b, feat + b, val + b, idx);
    update_linear_nz(m, &rv, tg[r]);
    update_bias_nz(m, tg[r]);
    if (m->model_type == FO_FFM) ffm_update_vector_nz(m, &rv, tg[r]);
    else if (m->model_type == FO_FM) fm_update_vector_nz(m, &rv, tg[r], svx + (size_t)r * k);
  }
  free(idx); free(tg); free(svx);
  return tmp_loss;
}


/* ---------------- block update by reductions (the engine's batch semantics) ----------------
 *
 * With w and tmp_grad frozen over the block, the touches t = 0..T-1 of ONE accumulator (n, z),
 * taken in row order, are folded like this.  The occurrences of a feature in the block (the rows of
 * the block for the bias), in row order, are cut into segments of FO_SEG; the touches an accumulator
 * receives from the occurrences of segment s form its segment s (possibly empty: a row may lack the
 * partner field) -- so all accumulators of a feature share the cut:
 *   P_s = sum of g_t*g_t, G_s = sum of g_t      (inside a segment left to right; the sums start
 *                                               from -0.0f, the identity of fp addition)
 *   B_0 = n_0, B_{s+1} = B_s + P_s,  n_T = B_S   (segment totals joined left to right)
 *   n_t = B_s + (the segment's partial sum of g*g before touch t)
 * and, with sigma the reference's step size (ftrl_model.cpp:71, ffm.cpp:113/:118, fm.cpp:92):
 *   - every touch "plain" (the square root sees n + g*g: linear, bias, FM, the FFM pair's first
 *     slot): the sigmas telescope, sum sigma_t = (sqrtf(n_T) - sqrtf(n_0)) / alpha  
 *   - from the first touch t0 of the ffm.cpp:118 kind (the square root sees n + g2*g1) on, every
 *     touch contributes its own difference of roots d_t = sqrtf(n_t + q_t) - sqrtf(n_t), summed
 *     per segment then over the segments (D); the plain touches before t0 telescope to
 *     sqrtf(n_t0) - sqrtf(n_0); the step sizes' sum is (that + D) / alpha -- ONE divide
 *   latent:        z_T = (z_0 + G) - ((sum of root differences) / alpha) * w
 *                                               (ffm.cpp:114/:119, fm.cpp:93: z + g - sigma*w)
 *   linear / bias: z_T = z_0 + (G - sigma*w)    (ftrl_model.cpp:72/:83: z += g - sigma*w)
 * One touch evaluates the reference's expression literally, so n_rows == 1 is fo_train bit for bit
 * -- as long as no accumulator is touched twice by one row.  Where a row does that (a field with
 * several entries: the partner slots see one touch per entry; the same id twice in a row) the
 * reference's running update inside the row is not a sum, and those accumulators ("serial") keep
 * the row walk above for the whole block.  Which ones: see mark_serial(). */
#define FO_SEG 16
int fo_block_segment(void) { return FO_SEG; }

typedef struct {
  float P, G, D;          /* the running segment: sum g*g, sum g, sum of root differences */
  float B, Gacc, Dacc;    /* the segments before it */
  float ncap;             /* n_t at the first :118 touch */
  int any, seen, head_plain;
} fo_acc;

static inline void acc_init(fo_acc *a, float n0) {
  a->P = a->G = a->D = a->Gacc = a->Dacc = -0.0f;
  a->B = n0;
  a->ncap = 0.0f;
  a->any = a->seen = a->head_plain = 0;
}
static inline void acc_flush(fo_acc *a) {
  a->B = a->B + a->P;
  a->Gacc = a->Gacc + a->G;
  a->Dacc = a->Dacc + a->D;
  a->P = a->G = a->D = -0.0f;
}
/* before the touches of the feature's occurrence number `occ` (0-based): a new segment starts at
 * every multiple of FO_SEG (joining an empty segment adds -0.0f: nothing) */
static inline void acc_at(fo_acc *a, int occ) {
  if (occ > 0 && occ % FO_SEG == 0) acc_flush(a);
}
/* one touch: gradient g, what the square root adds to n (q), plain = (q is g*g by construction) */
static inline void acc_touch(fo_acc *a, float g, float q, int plain) {
  const float nt = a->B + a->P;
  if (!a->any) { a->any = 1; a->head_plain = plain; }
  if (!plain && !a->seen) { a->seen = 1; a->ncap = nt; }
  if (a->seen) a->D = a->D + (sqrtf(nt + q) - sqrtf(nt));
  a->G = a->G + g;
  a->P = a->P + g * g;
}
static inline void acc_finish_latent(const fo_model *m, fo_acc *a, float w, float *n, float *z) {
  if (!a->any) return;
  const float n0 = *n;
  acc_flush(a);
  float S = -0.0f;
  if (a->head_plain) {
    const float ncap = a->seen ? a->ncap : a->B;
    S = S + (sqrtf(ncap) - sqrtf(n0));
  }
  S = S + a->Dacc;
  *z = (*z + a->Gacc) - (S / m->w_alpha) * w;
  *n = a->B;
}
static inline void acc_finish_linear(const fo_model *m, fo_acc *a, float w, float *n, float *z) {
  if (!a->any) return;
  const float n0 = *n;
  acc_flush(a);
  const float si = (sqrtf(a->B) - sqrtf(n0)) / m->w_alpha;
  *z = *z + (a->Gacc - si * w);
  *n = a->B;
}

typedef struct { int feat, p; } fo_ent;
static int ent_cmp(const void *x, const void *y) {
  const fo_ent *a = (const fo_ent *)x, *b = (const fo_ent *)y;
  if (a->feat != b->feat) return a->feat < b->feat ? -1 : 1;
  return a->p < b->p ? -1 : (a->p > b->p);
}

/* The block grouped by feature, as the engine's grouping produces it (kernels_group.h): the
 * surviving entries sorted by (feature, entry index), per row and field the count and the first
 * surviving entry. */
typedef struct {
  int n_ent;       /* surviving entries */
  fo_ent *ent;     /* sorted */
  int *row_of;     /* [nnz] */
  int *gid;        /* [nnz] start of the entry's group in ent[] (-1: erased) */
  int *rcnt, *rfirst; /* [n_rows * F] (FFM) */
  int F;
} fo_groups;

static void groups_build(const fo_model *m, fo_groups *g, int n_rows, const int32_t *row_ptr,
                         const int32_t *field, const int32_t *feat) {
  const int nnz = row_ptr[n_rows];
  g->F = m->model_type == FO_FFM ? m->n_fields : 1;
  g->ent = (fo_ent *)malloc(sizeof(fo_ent) * (size_t)(nnz > 0 ? nnz : 1));
  g->row_of = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  g->gid = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  const size_t rf = (size_t)(n_rows > 0 ? n_rows : 1) * (size_t)g->F;
  g->rcnt = (int *)calloc(rf, sizeof(int));
  g->rfirst = (int *)malloc(sizeof(int) * rf);
  for (size_t i = 0; i < rf; i++) g->rfirst[i] = -1;
  g->n_ent = 0;
  for (int r = 0; r < n_rows; r++)
    for (int p = row_ptr[r]; p < row_ptr[r + 1]; p++) {
      g->row_of[p] = r;
      g->gid[p] = -1;
      const int f = m->model_type == FO_FFM ? field[p] : 0;
      if (!in_range(m, f, feat[p])) continue;
      g->ent[g->n_ent].feat = feat[p];
      g->ent[g->n_ent].p = p;
      g->n_ent++;
      const size_t c = (size_t)r * g->F + (size_t)f;
      if (g->rcnt[c]++ == 0) g->rfirst[c] = p;
    }
  qsort(g->ent, (size_t)g->n_ent, sizeof(fo_ent), ent_cmp);
  for (int t = 0, lo = 0; t < g->n_ent; t++) {
    if (t > 0 && g->ent[t].feat != g->ent[t - 1].feat) lo = t;
    g->gid[g->ent[t].p] = lo;
  }
}
static void groups_free(fo_groups *g) {
  free(g->ent); free(g->row_of); free(g->gid); free(g->rcnt); free(g->rfirst);
}

/* Which accumulators of the group ent[lo, hi) keep the row walk.  Mirrors what the engine's
 * grouping can see (kernels_group.h: cmask):
 *   FFM: slot fp of the feature, when some row of the feature holds two or more surviving entries
 *        of field fp, or -- for every slot that row touches -- when the feature itself occurs twice
 *        in the row.  More than 64 fields (no field masks on the device): every slot.
 *   FM / LR: the whole feature, when it occurs twice in some row.
 * ser[fp] (FFM) / ser[0] (FM, LR) receive 0 / 1. */
static void mark_serial(const fo_model *m, const fo_groups *g, const int32_t *field, int lo, int hi,
                        unsigned char *ser) {
  const int F = g->F;
  memset(ser, 0, (size_t)F);
  for (int t = lo; t < hi; t++) {
    const int p = g->ent[t].p, r = g->row_of[p];
    const int dup = (t > lo && g->row_of[g->ent[t - 1].p] == r) ||
                    (t + 1 < hi && g->row_of[g->ent[t + 1].p] == r);
    if (m->model_type != FO_FFM) {
      if (dup) ser[0] = 1;
      continue;
    }
    if (F > 64) { memset(ser, 1, (size_t)F); return; }
    const int fa = field[p];
    for (int fp = 0; fp < F; fp++) {
      const int c = g->rcnt[(size_t)r * F + fp];
      if (c >= 2) ser[fp] = 1;
      if (dup && c - (fp == fa ? 1 : 0) >= 1) ser[fp] = 1;
    }
  }
}

double fo_train_batch(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                      const int32_t *feat, const float *val, const int32_t *label,
                      float *logit_out) {
  const int k = m->n_factors;
  const int64_t L = m->row_len;
  int max_nnz = 1;
  for (int r = 0; r < n_rows; r++)
    if (row_ptr[r + 1] - row_ptr[r] > max_nnz) max_nnz = row_ptr[r + 1] - row_ptr[r];
  int *idx = (int *)malloc(sizeof(int) * (size_t)max_nnz);
  float *tg = (float *)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1));
  float *svx = NULL;
  if (m->model_type == FO_FM)
    svx = (float *)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * (size_t)(k > 0 ? k : 1));
  rowview rv;
  double tmp_loss = 0.0;
  /* sweep 1: lazy refresh of everything the batch touches, from the batch-start (n,z) */
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r];
    rv_build(m, &rv, row_ptr[r + 1] - b, field + b, feat + b, val + b, idx);
    update_linear_w(m, &rv);
    if (m->model_type == FO_FFM) ffm_update_vector_w(m, &rv);
    else if (m->model_type == FO_FM) fm_update_vector_w(m, &rv);
  }
  if (n_rows > 0) update_bias(m);
  /* sweep 2: forward with frozen weights */
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r];
    rv_build(m, &rv, row_ptr[r + 1] - b, field + b, feat + b, val + b, idx);
    float logit;
    if (m->model_type == FO_FFM) logit = compute_ffm_logit(m, &rv);
    else if (m->model_type == FO_FM) logit = compute_fm_logit(m, &rv, svx + (size_t)r * k);
    else logit = compute_linear_logit(m, &rv);
    tg[r] = fo_sigmoid(logit) - (float)label[r];
    if (logit_out) logit_out[r] = logit;
    tmp_loss += fo_loss(label[r], logit);
  }
  /* sweep 3: the accumulators.  Bias: one plain touch per row, g = tmp_grad (ftrl_model.cpp:79-85) */
  if (n_rows > 0) {
    fo_acc a;
    acc_init(&a, m->bias3[1]);
    for (int r = 0; r < n_rows; r++) {
      acc_at(&a, r);
      acc_touch(&a, tg[r], tg[r] * tg[r], 1);
    }
    acc_finish_linear(m, &a, m->bias3[0], &m->bias3[1], &m->bias3[2]);
  }
  fo_groups g;
  groups_build(m, &g, n_rows, row_ptr, field, feat);
  const int F = g.F;
  /* per group: serial flags, kept per entry group for the row walk below */
  unsigned char *ser_all = (unsigned char *)calloc((size_t)(g.n_ent > 0 ? g.n_ent : 1) * (size_t)F, 1);
  int any_serial = 0;
  for (int lo = 0; lo < g.n_ent;) {
    int hi = lo + 1;
    while (hi < g.n_ent && g.ent[hi].feat == g.ent[lo].feat) hi++;
    unsigned char *ser = ser_all + (size_t)lo * F;
    mark_serial(m, &g, field, lo, hi, ser);
    for (int f = 0; f < F; f++) any_serial |= ser[f];
    const int i = g.ent[lo].feat;
    /* linear: g = tmp_grad * x (ftrl_model.cpp:66-77); serial when the id repeats inside a row */
    int lin_serial = 0;
    for (int t = lo + 1; t < hi; t++)
      if (g.row_of[g.ent[t].p] == g.row_of[g.ent[t - 1].p]) lin_serial = 1;
    if (lin_serial) {
      for (int t = lo; t < hi; t++) {
        const int p = g.ent[t].p;
        nz_step(m, m->lin_w[i], tg[g.row_of[p]] * val[p], &m->lin_n[i], &m->lin_z[i]);
      }
    } else {
      fo_acc a;
      acc_init(&a, m->lin_n[i]);
      for (int t = lo; t < hi; t++) {
        const int p = g.ent[t].p;
        const float gg = tg[g.row_of[p]] * val[p];
        acc_at(&a, t - lo);
        acc_touch(&a, gg, gg * gg, 1);
      }
      acc_finish_linear(m, &a, m->lin_w[i], &m->lin_n[i], &m->lin_z[i]);
    }
    if (m->model_type == FO_FM) {
      /* fm.cpp:84-95: g = tmp_grad * (x * s_vx - v * x * x), every touch plain */
      if (ser[0]) {
        for (int t = lo; t < hi; t++) {
          const int p = g.ent[t].p, r = g.row_of[p];
          fm_feat_step(m, i, val[p], tg[r], svx + (size_t)r * k);
        }
      } else {
        for (int f = 0; f < k; f++) {
          const int64_t o = (int64_t)i * k + f;
          const float w = m->vec_w[o];
          fo_acc a;
          acc_init(&a, m->vec_n[o]);
          for (int t = lo; t < hi; t++) {
            const int p = g.ent[t].p, r = g.row_of[p];
            const float x = val[p];
            const float gg = tg[r] * (x * svx[(size_t)r * k + f] - w * x * x);
            acc_at(&a, t - lo);
            acc_touch(&a, gg, gg * gg, 1);
          }
          acc_finish_latent(m, &a, w, &m->vec_n[o], &m->vec_z[o]);
        }
      }
    } else if (m->model_type == FO_FFM) {
      /* ffm.cpp:102-121 slot by slot: slot (i, fp) is touched by every occurrence of i whose row
       * holds an entry q of field fp other than itself (exactly one here: the slot is not serial) */
      for (int fp = 0; fp < F; fp++) {
        if (ser[fp]) continue;
        for (int f = 0; f < k; f++) {
          const int64_t o = i * L + (int64_t)fp * k + f;
          const float w = m->vec_w[o];
          fo_acc a;
          acc_init(&a, m->vec_n[o]);
          for (int t = lo; t < hi; t++) {
            const int p = g.ent[t].p, r = g.row_of[p];
            const int q = g.rfirst[(size_t)r * F + fp];
            acc_at(&a, t - lo);
            if (q < 0 || q == p) continue;
            const float x = p < q ? val[p] * val[q] : val[q] * val[p];
            const float vp = m->vec_w[feat[q] * L + (int64_t)field[p] * k + f];
            const float gg = tg[r] * vp * x;
            if (p < q || m->learn) {
              acc_touch(&a, gg, gg * gg, 1); /* ffm.cpp:112-115 */
            } else {
              const float g1 = tg[r] * w * x;      /* the pair's first entry's gradient */
              acc_touch(&a, gg, gg * g1, 0); /* ffm.cpp:117-120 incl. :118 */
            }
          }
          acc_finish_latent(m, &a, w, &m->vec_n[o], &m->vec_z[o]);
        }
      }
    }
    lo = hi;
  }
  /* the serial FFM slots: the row walk of ffm.cpp:90-136, restricted to them */
  if (m->model_type == FO_FFM && any_serial) {
    for (int r = 0; r < n_rows; r++) {
      const int b = row_ptr[r];
      rv_build(m, &rv, row_ptr[r + 1] - b, field + b, feat + b, val + b, idx);
      for (int a = 0; a < rv.n; a++)
        for (int c = a + 1; c < rv.n; c++) {
          const int pa = b + rv.idx[a], pc = b + rv.idx[c];
          const int field1 = field[pa], i = feat[pa], field2 = field[pc], j = feat[pc];
          const int s1 = ser_all[(size_t)g.gid[pa] * F + field2], s2 = ser_all[(size_t)g.gid[pc] * F + field1];
          if (!s1 && !s2) continue;
          const float x = val[pa] * val[pc];
          const int64_t o1 = i * L + (int64_t)field2 * k, o2 = j * L + (int64_t)field1 * k;
          for (int f = 0; f < k; f++) {
            const float v1 = m->vec_w[o1 + f], v2 = m->vec_w[o2 + f];
            float n1 = m->vec_n[o1 + f], z1 = m->vec_z[o1 + f];
            if (o1 == o2) {
              /* the same (field, id) twice in the row: ONE slot on both sides of the pair (the
               * reference deadlocks here, see fo_train_batch_rowwalk): its i-side step, then its
               * j-side step on the result; g1 == g2, both partner weights are this slot's */
              const float gg = tg[r] * v1 * x;
              const float sa = (sqrtf(n1 + gg * gg) - sqrtf(n1)) / m->w_alpha;
              z1 = z1 + gg - sa * v1;
              n1 = n1 + gg * gg;
              const float sb = (sqrtf(n1 + gg * gg) - sqrtf(n1)) / m->w_alpha;
              z1 = z1 + gg - sb * v1;
              n1 = n1 + gg * gg;
              if (s1) { m->vec_n[o1 + f] = n1; m->vec_z[o1 + f] = z1; }
              continue;
            }
            float n2 = m->vec_n[o2 + f], z2 = m->vec_z[o2 + f];
            ffm_pair_step(m, tg[r], x, v1, v2, &n1, &z1, &n2, &z2);
            if (s1) { m->vec_n[o1 + f] = n1; m->vec_z[o1 + f] = z1; }
            if (s2) { m->vec_n[o2 + f] = n2; m->vec_z[o2 + f] = z2; }
          }
        }
    }
  }
  groups_free(&g);
  free(ser_all);
  free(idx); free(tg); free(svx);
  return tmp_loss;
}

/* ---------------- reference-style threaded epoch (CPU baseline) ---------------- */

typedef struct {
  fo_model *m;
  int r0, r1;
  const int32_t *row_ptr, *field, *feat;
  const float *val;
  const int32_t *label;
  double loss;
} worker_arg;

static inline void lock2(fo_model *m, int i, int j) {
  if (i == j) { pthread_mutex_lock(&m->locks[i]); return; }
  if (i > j) { int t = i; i = j; j = t; }
  pthread_mutex_lock(&m->locks[i]);
  pthread_mutex_lock(&m->locks[j]);
}
static inline void unlock2(fo_model *m, int i, int j) {
  pthread_mutex_unlock(&m->locks[i]);
  if (i != j) pthread_mutex_unlock(&m->locks[j]);
}

/* train() with the reference's locking structure; forward reads are lock-free as there. */
static float train_locked(fo_model *m, int nnz, const int32_t *field, const int32_t *feat,
                          const float *val, int label, float *sum_vx) {
  int stack_idx[FO_STACK_NNZ];
  if (nnz > FO_STACK_NNZ) nnz = FO_STACK_NNZ;
  rowview rv;
  rv_build(m, &rv, nnz, field, feat, val, stack_idx);
  const int k = m->n_factors;
  const int64_t L = m->row_len;
  for (int a = 0; a < rv.n; a++) { /* ftrl_model.cpp:52-59 */
    const int i = RV_FEAT(&rv, a);
    pthread_mutex_lock(&m->locks[i]);
    m->lin_w[i] = mzw(m, m->lin_n[i], m->lin_z[i]);
    pthread_mutex_unlock(&m->locks[i]);
  }
  pthread_mutex_lock(&m->bias_lock);
  update_bias(m);
  pthread_mutex_unlock(&m->bias_lock);
  float logit;
  if (m->model_type == FO_FFM) {
    for (int a = 0; a < rv.n; a++) /* ffm.cpp:72-88 */
      for (int b = a + 1; b < rv.n; b++) {
        const int field1 = RV_FIELD(&rv, a), i = RV_FEAT(&rv, a);
        const int field2 = RV_FIELD(&rv, b), j = RV_FEAT(&rv, b);
        lock2(m, i, j);
        for (int f = 0; f < k; f++) {
          const int64_t f1 = i * L + (int64_t)field2 * k + f;
          m->vec_w[f1] = mzw_latent(m, m->vec_n[f1], m->vec_z[f1], m->vec_w[f1]);
          const int64_t f2 = j * L + (int64_t)field1 * k + f;
          m->vec_w[f2] = mzw_latent(m, m->vec_n[f2], m->vec_z[f2], m->vec_w[f2]);
        }
        unlock2(m, i, j);
      }
    logit = compute_ffm_logit(m, &rv);
  } else if (m->model_type == FO_FM) {
    for (int a = 0; a < rv.n; a++) { /* fm.cpp:69-78 */
      const int i = RV_FEAT(&rv, a);
      pthread_mutex_lock(&m->locks[i]);
      for (int f = 0; f < k; f++) {
        const int64_t o = (int64_t)i * k + f;
        m->vec_w[o] = mzw_latent(m, m->vec_n[o], m->vec_z[o], m->vec_w[o]);
      }
      pthread_mutex_unlock(&m->locks[i]);
    }
    logit = compute_fm_logit(m, &rv, sum_vx);
  } else {
    logit = compute_linear_logit(m, &rv);
  }
  const float tmp_grad = fo_sigmoid(logit) - (float)label;
  for (int a = 0; a < rv.n; a++) { /* ftrl_model.cpp:66-77 */
    const int i = RV_FEAT(&rv, a);
    pthread_mutex_lock(&m->locks[i]);
    nz_step(m, m->lin_w[i], tmp_grad * RV_VAL(&rv, a), &m->lin_n[i], &m->lin_z[i]);
    pthread_mutex_unlock(&m->locks[i]);
  }
  pthread_mutex_lock(&m->bias_lock);
  update_bias_nz(m, tmp_grad);
  pthread_mutex_unlock(&m->bias_lock);
  if (m->model_type == FO_FFM) {
    float tn1[256], tz1[256], tn2[256], tz2[256];
    for (int a = 0; a < rv.n; a++) /* ffm.cpp:90-136: read under lock, copy back under lock */
      for (int b = a + 1; b < rv.n; b++) {
        const int field1 = RV_FIELD(&rv, a), i = RV_FEAT(&rv, a);
        const int field2 = RV_FIELD(&rv, b), j = RV_FEAT(&rv, b);
        const float x = RV_VAL(&rv, a) * RV_VAL(&rv, b);
        const int64_t o1 = i * L + (int64_t)field2 * k, o2 = j * L + (int64_t)field1 * k;
        const int kk = k <= 256 ? k : 256;
        lock2(m, i, j);
        for (int f = 0; f < kk; f++) {
          tn1[f] = m->vec_n[o1 + f]; tz1[f] = m->vec_z[o1 + f];
          tn2[f] = m->vec_n[o2 + f]; tz2[f] = m->vec_z[o2 + f];
          ffm_pair_step(m, tmp_grad, x, m->vec_w[o1 + f], m->vec_w[o2 + f], &tn1[f], &tz1[f],
                        &tn2[f], &tz2[f]);
        }
        unlock2(m, i, j);
        lock2(m, i, j);
        for (int f = 0; f < kk; f++) { m->vec_z[o1 + f] = tz1[f]; m->vec_n[o1 + f] = tn1[f]; }
        for (int f = 0; f < kk; f++) { m->vec_z[o2 + f] = tz2[f]; m->vec_n[o2 + f] = tn2[f]; }
        unlock2(m, i, j);
      }
  } else if (m->model_type == FO_FM) {
    for (int a = 0; a < rv.n; a++) { /* fm.cpp:80-101 */
      const int i = RV_FEAT(&rv, a);
      pthread_mutex_lock(&m->locks[i]);
      fm_feat_step(m, i, RV_VAL(&rv, a), tmp_grad, sum_vx);
      pthread_mutex_unlock(&m->locks[i]);
    }
  }
  return logit;
}

static void *worker(void *p) {
  worker_arg *w = (worker_arg *)p;
  float *sum_vx = (float *)malloc(sizeof(float) * (size_t)(w->m->n_factors > 0 ? w->m->n_factors : 1));
  double tmp_loss = 0.0;
  for (int r = w->r0; r < w->r1; r++) { /* ftrl_offline.cpp:74-83 */
    const int b = w->row_ptr[r], e = w->row_ptr[r + 1];
    const float logit = train_locked(w->m, e - b, w->field + b, w->feat + b, w->val + b,
                                     w->label[r], sum_vx);
    tmp_loss += fo_loss(w->label[r], logit);
  }
  w->loss = tmp_loss;
  free(sum_vx);
  return NULL;
}

double fo_train_rows_threaded(fo_model *m, int n_threads, int n_rows, const int32_t *row_ptr,
                              const int32_t *field, const int32_t *feat, const float *val,
                              const int32_t *label, double *loss_sum) {
  if (n_threads < 1) n_threads = 1;
  if (!m->locks) {
    m->locks = (pthread_mutex_t *)malloc(sizeof(pthread_mutex_t) * (size_t)(m->n_feats > 0 ? m->n_feats : 1));
    for (int i = 0; i < m->n_feats; i++) pthread_mutex_init(&m->locks[i], NULL);
  }
  worker_arg *args = (worker_arg *)calloc((size_t)n_threads, sizeof(worker_arg));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  const int unit = (n_rows + n_threads - 1) / n_threads; /* ftrl_offline.cpp:65 */
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < n_threads; t++) {
    args[t].m = m;
    args[t].r0 = t * unit < n_rows ? t * unit : n_rows;
    args[t].r1 = (t + 1) * unit < n_rows ? (t + 1) * unit : n_rows;
    args[t].row_ptr = row_ptr; args[t].field = field; args[t].feat = feat;
    args[t].val = val; args[t].label = label;
    pthread_create(&th[t], NULL, worker, &args[t]);
  }
  double total = 0.0;
  for (int t = 0; t < n_threads; t++) { pthread_join(th[t], NULL); total += args[t].loss; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (loss_sum) *loss_sum = total;
  free(args); free(th);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
