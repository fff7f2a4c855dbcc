// engine_state.h -- weights and optimizer state in and out of the engine: dense arrays in the
// reference's save order <-> the stored records (get/set_weights, get/set_state, get/set_rows).
// Part of engine.hip's translation unit (included inside its extern "C" block).

// ---- dense <-> record layout transfers ----------------------------------------------------

static int vec_transfer(ffm_engine *e, int comp, float *host, bool to_host) {
  if (!host || e->logical_len == 0) return FFM_OK;
  const int64_t RL = e->logical_len;
  const int64_t chunk = e->stage_floats / RL;
  for (int64_t f0 = 0; f0 < e->m.n_feats; f0 += chunk) {
    const int64_t nf = std::min<int64_t>(chunk, e->m.n_feats - f0);
    const size_t bytes = static_cast<size_t>(nf * RL) * sizeof(float);
    if (!to_host) HIP_TRY(hipMemcpyAsync(e->d_stage, host + f0 * RL, bytes, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(lat_component_copy_kernel, dim3(1024), dim3(256), 0, e->stream, e->m,
                       static_cast<int>(RL), comp, e->d_stage, f0, nf, to_host ? 1 : 0);
    if (to_host) HIP_TRY(hipMemcpyAsync(host + f0 * RL, e->d_stage, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return FFM_OK;
}

static int flat_transfer(ffm_engine *e, float *dev, float *host, size_t n, bool to_host) {
  if (!host) return FFM_OK;
  HIP_TRY(hipMemcpyAsync(to_host ? host : dev, to_host ? dev : host, n * sizeof(float),
                         to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return FFM_OK;
}

int ffm_engine_set_weights(ffm_engine *e, const float *bias, const float *lin_w, const float *vec_w) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (int rc_e = eval_launch_pending(e)) return rc_e;  // (a deferred evaluation block sees the state as it was)
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  int rc;
  if ((rc = flat_transfer(e, e->m.bias3 + 0, const_cast<float *>(bias), 1, false))) return rc;
  if ((rc = flat_transfer(e, e->m.lin_w, const_cast<float *>(lin_w), e->m.n_feats, false))) return rc;
  return vec_transfer(e, LAT_W, const_cast<float *>(vec_w), false);
}

int ffm_engine_get_weights(ffm_engine *e, float *bias, float *lin_w, float *vec_w) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (int rc_e = eval_launch_pending(e)) return rc_e;  // (a deferred evaluation block sees the state as it was)
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  int rc;
  if ((rc = flat_transfer(e, e->m.bias3 + 0, bias, 1, true))) return rc;
  if ((rc = flat_transfer(e, e->m.lin_w, lin_w, e->m.n_feats, true))) return rc;
  return vec_transfer(e, LAT_W, vec_w, true);
}

int ffm_engine_set_state(ffm_engine *e, const float *bias_n, const float *bias_z,
                         const float *lin_n, const float *lin_z, const float *vec_n,
                         const float *vec_z) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (int rc_e = eval_launch_pending(e)) return rc_e;  // (a deferred evaluation block sees the state as it was)
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  int rc;
  if ((rc = flat_transfer(e, e->m.bias3 + 1, const_cast<float *>(bias_n), 1, false))) return rc;
  if ((rc = flat_transfer(e, e->m.bias3 + 2, const_cast<float *>(bias_z), 1, false))) return rc;
  if ((rc = flat_transfer(e, e->m.lin_n, const_cast<float *>(lin_n), e->m.n_feats, false))) return rc;
  if ((rc = flat_transfer(e, e->m.lin_z, const_cast<float *>(lin_z), e->m.n_feats, false))) return rc;
  if ((rc = vec_transfer(e, LAT_N, const_cast<float *>(vec_n), false))) return rc;
  return vec_transfer(e, LAT_Z, const_cast<float *>(vec_z), false);
}

int ffm_engine_get_state(ffm_engine *e, float *bias_n, float *bias_z, float *lin_n, float *lin_z,
                         float *vec_n, float *vec_z) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (int rc_e = eval_launch_pending(e)) return rc_e;  // (a deferred evaluation block sees the state as it was)
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  int rc;
  if ((rc = flat_transfer(e, e->m.bias3 + 1, bias_n, 1, true))) return rc;
  if ((rc = flat_transfer(e, e->m.bias3 + 2, bias_z, 1, true))) return rc;
  if ((rc = flat_transfer(e, e->m.lin_n, lin_n, e->m.n_feats, true))) return rc;
  if ((rc = flat_transfer(e, e->m.lin_z, lin_z, e->m.n_feats, true))) return rc;
  if ((rc = vec_transfer(e, LAT_N, vec_n, true))) return rc;
  return vec_transfer(e, LAT_Z, vec_z, true);
}

// Gather / scatter of the records of a list of features (host arrays).
static int rows_transfer(ffm_engine *e, int32_t n, const int32_t *ids, float *const lin[3],
                         float *const vec[3], bool to_host) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (n < 0 || (n > 0 && !ids)) return fail(FFM_E_INVALID, "bad feature id list");
  for (int32_t j = 0; j < n; j++)
    if (ids[j] < 0 || ids[j] >= e->m.n_feats) return fail(FFM_E_INVALID, "feature id out of range");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  if (int rc_e = eval_launch_pending(e)) return rc_e;  // (a deferred evaluation block sees the state as it was)
  const int64_t RL = e->logical_len;
  const int64_t per = std::min<int64_t>(ffm_engine::kIdsCap, RL > 0 ? e->stage_floats / RL : ffm_engine::kIdsCap);
  float *const lin_dev[3] = {e->m.lin_n, e->m.lin_z, e->m.lin_w};
  for (int64_t j0 = 0; j0 < n; j0 += per) {
    const int nf = static_cast<int>(std::min<int64_t>(per, n - j0));
    HIP_TRY(hipMemcpyAsync(e->d_ids, ids + j0, sizeof(int) * nf, hipMemcpyHostToDevice, e->stream));
    for (int comp = 0; comp < 3; comp++) {
      if (lin[comp]) {
        if (!to_host) HIP_TRY(hipMemcpyAsync(e->d_stage, lin[comp] + j0, sizeof(float) * nf, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(lin_rows_copy_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, e->stream,
                           lin_dev[comp], e->m.n_feats, e->d_stage, e->d_ids, nf, to_host ? 1 : 0);
        if (to_host) HIP_TRY(hipMemcpyAsync(lin[comp] + j0, e->d_stage, sizeof(float) * nf, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
      }
      if (vec[comp] && RL > 0) {
        const size_t bytes = static_cast<size_t>(nf) * RL * sizeof(float);
        if (!to_host) HIP_TRY(hipMemcpyAsync(e->d_stage, vec[comp] + j0 * RL, bytes, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(lat_rows_copy_kernel, dim3(1024), dim3(256), 0, e->stream, e->m,
                           static_cast<int>(RL), comp, e->d_stage, e->d_ids,
                           static_cast<int64_t>(nf), to_host ? 1 : 0);
        if (to_host) HIP_TRY(hipMemcpyAsync(vec[comp] + j0 * RL, e->d_stage, bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
      }
    }
  }
  HIP_TRY(hipGetLastError());
  return FFM_OK;
}

int ffm_engine_get_rows(ffm_engine *e, int32_t n, const int32_t *feat_ids, float *lin_w,
                        float *lin_n, float *lin_z, float *vec_w, float *vec_n, float *vec_z) {
  float *const lin[3] = {lin_n, lin_z, lin_w}, *const vec[3] = {vec_n, vec_z, vec_w};  // LAT_* order
  return rows_transfer(e, n, feat_ids, lin, vec, true);
}

int ffm_engine_set_rows(ffm_engine *e, int32_t n, const int32_t *feat_ids, const float *lin_w,
                        const float *lin_n, const float *lin_z, const float *vec_w,
                        const float *vec_n, const float *vec_z) {
  float *const lin[3] = {const_cast<float *>(lin_n), const_cast<float *>(lin_z), const_cast<float *>(lin_w)};
  float *const vec[3] = {const_cast<float *>(vec_n), const_cast<float *>(vec_z), const_cast<float *>(vec_w)};
  return rows_transfer(e, n, feat_ids, lin, vec, false);
}
