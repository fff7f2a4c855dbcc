// engine_profile.h -- per-kernel HIP-event timing behind ffm_engine_profile_* (bench.py's table).
// Part of engine.hip's translation unit (included inside its extern "C" block).

// ---- profiling -----------------------------------------------------------------------------

int ffm_engine_profile_enable(ffm_engine *e, int32_t on) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (int rc_w = e->drain()) return rc_w;
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  HIP_TRY(hipDeviceSynchronize());
  for (auto &r : e->prof) { e->event_pool.push_back(r.e0); e->event_pool.push_back(r.e1); }
  e->prof.clear();
  e->prof_on = on != 0;
  e->prof_only = -1;
  return FFM_OK;
}

static int profile_totals(ffm_engine *e, double *ms, int *n);

// The kernel with the largest total time among those on the step's critical path.  The look-ahead
// grouping is left out: it runs beside the step on its own queue, and its "time" is mostly waiting.
static int dominant_kernel(const double *ms) {
  int best = K_ROW;
  for (int k = 0; k < K_COUNT; k++) {
    if (k == K_GROUP_KEYS || k == K_GROUP_SORT || k == K_GROUP_FINISH) continue;
    if (ms[k] > ms[best]) best = k;
  }
  return best;
}

int ffm_engine_profile_focus(ffm_engine *e) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  double ms[K_COUNT];
  int n[K_COUNT];
  int rc = profile_totals(e, ms, n);
  if (rc) return rc;
  const int best = dominant_kernel(ms);
  for (auto &r : e->prof) { e->event_pool.push_back(r.e0); e->event_pool.push_back(r.e1); }
  e->prof.clear();
  e->prof_only = best;
  return FFM_OK;
}

static int profile_totals(ffm_engine *e, double *ms, int *n) {
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  HIP_TRY(hipDeviceSynchronize());
  for (int k = 0; k < K_COUNT; k++) { ms[k] = 0.0; n[k] = 0; }
  for (auto &r : e->prof) {
    float t = 0.0f;
    HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
    ms[r.kid] += t;
    n[r.kid]++;
  }
  return FFM_OK;
}

int ffm_engine_profile_read(ffm_engine *e, int32_t *n_launches, double *total_ms,
                            char *kernel_name, size_t kernel_name_cap) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  double ms[K_COUNT];
  int n[K_COUNT];
  int rc = profile_totals(e, ms, n);
  if (rc) return rc;
  const int best = dominant_kernel(ms);
  if (n_launches) *n_launches = n[best];
  if (total_ms) *total_ms = ms[best];
  if (kernel_name && kernel_name_cap) {
    std::string name = kKernelNames[best];
    if (best == K_REFRESH) name = "ffm_refresh_kernel";
    else if (best == K_LATENT_UPDATE_SINGLE) name = "ffm_update_single_kernel";
    else if (best == K_LATENT_UPDATE_FEW) name = "ffm_update_small_flat_kernel";
    else if (best == K_LATENT_UPDATE)
      name = e->m.type == FFM_MODEL_FM ? "fm_update_kernel"
             : e->m.n_factors % 4 == 0 && e->m.n_fields <= 64 ? "ffm_update_all_kernel" : "ffm_update_generic_kernel";
    else if (best == K_ROW || best == K_PREDICT_ROW)
      name = std::string(e->m.type == FFM_MODEL_FM ? "fm_" : "ffm_") +
             (best == K_ROW ? "row_kernel<train>" : "row_kernel<predict>");
    std::snprintf(kernel_name, kernel_name_cap, "%s", name.c_str());
  }
  return FFM_OK;
}

// Text table of every kernel's launches and total time since profiling was enabled.
int ffm_engine_profile_dump(ffm_engine *e, char *buf, size_t cap) {
  if (!e || !buf || !cap) return fail(FFM_E_INVALID, "null argument");
  double ms[K_COUNT];
  int n[K_COUNT];
  int rc = profile_totals(e, ms, n);
  if (rc) return rc;
  std::string out;
  char line[160];
  for (int k = 0; k < K_COUNT; k++) {
    if (!n[k]) continue;
    std::snprintf(line, sizeof line, "%-24s launches=%6d total_ms=%10.3f avg_us=%10.2f\n",
                  kKernelNames[k], n[k], ms[k], 1000.0 * ms[k] / n[k]);
    out += line;
  }
  std::snprintf(buf, cap, "%s", out.c_str());
  return FFM_OK;
}
