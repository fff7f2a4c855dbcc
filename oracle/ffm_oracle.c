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

/* Mini-batch semantics (see header).  Three sweeps over the batch; each sweep visits rows in
 * order and, inside a row, follows the reference's statement order. */
double fo_train_batch(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
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
