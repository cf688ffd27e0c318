// pf_host.h -- host side of the particle-filter entry points (auvp_pf_*, include/auvplan.h).
// Included at the end of auvplan.hip.
#ifndef AUVP_PF_HOST_H
#define AUVP_PF_HOST_H

// the kernels live in pf_kernels.hip (a translation unit with its own compiler flags); these are its launchers
extern "C" {
int auvpi_pf_threads(void);
size_t auvpi_pf_lds_bytes(int N);
hipError_t auvpi_pf_create_launch(const auvp::PfDev* D, hipStream_t stream);
hipError_t auvpi_pf_step_launch(const auvp::PfDev* D, hipStream_t stream);
}

namespace {

struct PfState {
  bool ready = false;
  int F = 0, N = 0, S = 0, A = 0;
  bool have_log = false;
  DevBuf st, ent, llen, mt, mtpos, shark0, meas, shark, mean, err, out_len, status, ndraw, updated, choice;
};

PfState* pf_of(auvp_handle* h) {
  if (!h->pf) {
    h->pf = new PfState();
    h->pf_free = [](void* p) { delete static_cast<PfState*>(p); };
  }
  return static_cast<PfState*>(h->pf);
}

const int PF_MAX_N = 2048;

int pf_alloc(auvp_handle* h, PfState& P, int F, int N) {
  if (F <= 0 || N <= 0) return fail(h, AUVP_ERR_ARG, "need F > 0 filters and N > 0 particles");
  if (N > PF_MAX_N) return fail(h, AUVP_ERR_CAPACITY, "%d particles per filter > %d (one workgroup's LDS)", N, PF_MAX_N);
  P.ready = false;
  P.F = F; P.N = N;
  HIPCHK(h, P.st.reserve((size_t)F * 5 * N * sizeof(double)));
  HIPCHK(h, P.ent.reserve((size_t)F * N * sizeof(int32_t)));
  HIPCHK(h, P.llen.reserve((size_t)F * sizeof(int32_t)));
  HIPCHK(h, P.mt.reserve((size_t)F * 624 * sizeof(uint32_t)));
  HIPCHK(h, P.mtpos.reserve((size_t)F * sizeof(int32_t)));
  HIPCHK(h, P.status.reserve((size_t)F * sizeof(int32_t)));
  HIPCHK(h, P.ndraw.reserve((size_t)F * sizeof(uint64_t)));
  HIPCHK(h, hipMemsetAsync(P.status.p, 0, (size_t)F * sizeof(int32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(P.ndraw.p, 0, (size_t)F * sizeof(uint64_t), h->stream));
  return AUVP_OK;
}

int pf_set_rng(auvp_handle* h, PfState& P, const uint32_t* mt, const int32_t* pos) {
  for (int f = 0; f < P.F; f++)
    if (pos[f] < 0 || pos[f] > 624) return fail(h, AUVP_ERR_ARG, "filter %d: MT19937 position %d outside 0..624", f, pos[f]);
  int rc;
  if ((rc = upload(h, P.mt, mt, (size_t)P.F * 624))) return rc;
  return upload(h, P.mtpos, pos, (size_t)P.F);
}

auvp::PfDev pf_dev(PfState& P) {
  auvp::PfDev D{};
  D.F = P.F; D.N = P.N; D.A = P.A; D.S = P.S;
  D.st = P.st.as<double>(); D.ent = P.ent.as<int32_t>(); D.llen = P.llen.as<int32_t>();
  D.mt = P.mt.as<uint32_t>(); D.mtpos = P.mtpos.as<int32_t>();
  D.shark0 = P.shark0.as<double>(); D.meas = P.meas.as<double>(); D.shark = P.shark.as<double>();
  D.mean = P.mean.as<double>(); D.err = P.err.as<double>(); D.out_len = P.out_len.as<int32_t>();
  D.status = P.status.as<int32_t>(); D.ndraw = P.ndraw.as<unsigned long long>();
  return D;
}

}  // namespace

extern "C" {

int auvp_pf_create_batch(auvp_handle* h, int32_t F, int32_t N, const double* shark_xy0, const uint32_t* mt,
                         const int32_t* mt_pos) {
  if (!h || !shark_xy0 || !mt || !mt_pos) return h ? fail(h, AUVP_ERR_ARG, "null argument") : AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  PfState& P = *pf_of(h);
  int rc;
  if ((rc = pf_alloc(h, P, F, N))) return rc;
  if ((rc = pf_set_rng(h, P, mt, mt_pos))) return rc;
  if ((rc = upload(h, P.shark0, shark_xy0, (size_t)F * 2))) return rc;
  P.S = 0; P.A = 0;
  auvp::PfDev D = pf_dev(P);
  const size_t lds = auvpi_pf_lds_bytes(N);
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  HIPCHK(h, auvpi_pf_create_launch(&D, h->stream));
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms; h->last_grid = F; h->last_block = 256; h->last_lds = (int)lds;
  P.ready = true;
  return AUVP_OK;
}

int auvp_pf_set_particles(auvp_handle* h, int32_t F, int32_t N, const double* particles, const int32_t* obj,
                          const int32_t* list_len, const uint32_t* mt, const int32_t* mt_pos) {
  if (!h || !particles || !mt || !mt_pos) return h ? fail(h, AUVP_ERR_ARG, "null argument") : AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  PfState& P = *pf_of(h);
  int rc;
  if ((rc = pf_alloc(h, P, F, N))) return rc;
  if ((rc = pf_set_rng(h, P, mt, mt_pos))) return rc;
  std::vector<double> soa((size_t)F * 5 * N);
  std::vector<int32_t> ent((size_t)F * N), ll(F);
  for (int f = 0; f < F; f++) {
    int mx = N;
    for (int p = 0; p < N; p++) {
      for (int c = 0; c < 5; c++) soa[((size_t)f * 5 + c) * N + p] = particles[((size_t)f * N + p) * 5 + c];
      const int id = obj ? obj[(size_t)f * N + p] : p;
      if (id < 0 || id >= 5 * N) return fail(h, AUVP_ERR_ARG, "filter %d position %d: object id %d outside 0..%d", f, p, id, 5 * N - 1);
      ent[(size_t)f * N + p] = id;
      if (id + 1 > mx) mx = id + 1;
    }
    ll[f] = list_len ? std::max(list_len[f], mx) : mx;
    if (ll[f] > 5 * N) return fail(h, AUVP_ERR_ARG, "filter %d: list length %d > 5 N", f, ll[f]);
  }
  if ((rc = upload(h, P.st, soa.data(), soa.size()))) return rc;
  if ((rc = upload(h, P.ent, ent.data(), ent.size()))) return rc;
  if ((rc = upload(h, P.llen, ll.data(), ll.size()))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  P.S = 0; P.A = 0;
  P.ready = true;
  return AUVP_OK;
}

int auvp_pf_set_rng(auvp_handle* h, const uint32_t* mt, const int32_t* mt_pos) {
  if (!h || !mt || !mt_pos) return h ? fail(h, AUVP_ERR_ARG, "null argument") : AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready) return fail(h, AUVP_ERR_STATE, "no particle-filter batch");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = pf_set_rng(h, P, mt, mt_pos);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_pf_run(auvp_handle* h, int32_t n_steps, int32_t n_auv, int32_t phases, const double* meas, const double* shark_xy,
                int32_t flags) {
  if (!h) return AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready) return fail(h, AUVP_ERR_STATE, "no particle-filter batch");
  if (n_steps <= 0 || (phases & ~7) || !phases) return fail(h, AUVP_ERR_ARG, "bad n_steps / phases");
  if ((phases & auvp::PF_PHASE_WEIGHTS) && (n_auv <= 0 || !meas)) return fail(h, AUVP_ERR_ARG, "update_weights needs >= 1 measurement row");
  if ((phases & auvp::PF_PHASE_MEAN) && !shark_xy) return fail(h, AUVP_ERR_ARG, "meanError needs the shark position");
  HIPCHK(h, hipSetDevice(h->device));
  const int F = P.F, N = P.N, S = n_steps, A = (phases & auvp::PF_PHASE_WEIGHTS) ? n_auv : 0;
  P.S = S; P.A = A;
  int rc;
  if (A && (rc = upload(h, P.meas, meas, (size_t)S * F * A * 5))) return rc;
  if ((phases & auvp::PF_PHASE_MEAN) && (rc = upload(h, P.shark, shark_xy, (size_t)S * F * 2))) return rc;
  HIPCHK(h, P.mean.reserve((size_t)S * F * 2 * sizeof(double)));
  HIPCHK(h, P.err.reserve((size_t)S * F * sizeof(double)));
  HIPCHK(h, P.out_len.reserve((size_t)S * F * sizeof(int32_t)));
  HIPCHK(h, hipMemsetAsync(P.mean.p, 0, (size_t)S * F * 2 * sizeof(double), h->stream));
  HIPCHK(h, hipMemsetAsync(P.err.p, 0, (size_t)S * F * sizeof(double), h->stream));
  HIPCHK(h, hipMemsetAsync(P.out_len.p, 0, (size_t)S * F * sizeof(int32_t), h->stream));
  auvp::PfDev D = pf_dev(P);
  D.phases = phases;
  P.have_log = (flags & AUVP_FLAG_ITER_LOG) != 0;
  if (P.have_log) {
    HIPCHK(h, P.updated.reserve((size_t)S * F * N * 5 * sizeof(double)));
    HIPCHK(h, P.choice.reserve((size_t)S * F * N * sizeof(int32_t)));
    HIPCHK(h, hipMemsetAsync(P.updated.p, 0, (size_t)S * F * N * 5 * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(P.choice.p, 0, (size_t)S * F * N * sizeof(int32_t), h->stream));
    D.updated = P.updated.as<double>();
    D.choice = P.choice.as<int32_t>();
  }
  const size_t lds = auvpi_pf_lds_bytes(N);
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  HIPCHK(h, auvpi_pf_step_launch(&D, h->stream));  // (threads per filter: pf_kernels.hip)
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms; h->last_grid = F; h->last_block = auvpi_pf_threads(); h->last_lds = (int)lds;
  return AUVP_OK;
}

int auvp_pf_particles(auvp_handle* h, double* out, int32_t* obj) {
  if (!h) return AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready) return fail(h, AUVP_ERR_STATE, "no particle-filter batch");
  HIPCHK(h, hipSetDevice(h->device));
  const int F = P.F, N = P.N;
  if (out) {
    std::vector<double> soa((size_t)F * 5 * N);
    HIPCHK(h, hipMemcpy(soa.data(), P.st.p, soa.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int f = 0; f < F; f++)
      for (int p = 0; p < N; p++)
        for (int c = 0; c < 5; c++) out[((size_t)f * N + p) * 5 + c] = soa[((size_t)f * 5 + c) * N + p];
  }
  if (obj) HIPCHK(h, hipMemcpy(obj, P.ent.p, (size_t)F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_pf_particles_dev(auvp_handle* h, double** st_soa, int32_t** obj) {
  if (!h) return AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready) return fail(h, AUVP_ERR_STATE, "no particle-filter batch");
  if (st_soa) *st_soa = P.st.as<double>();
  if (obj) *obj = P.ent.as<int32_t>();
  return AUVP_OK;
}

int auvp_pf_estimates(auvp_handle* h, double* mean, double* range_error, int32_t* list_len) {
  if (!h) return AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready || P.S <= 0) return fail(h, AUVP_ERR_STATE, "no particle-filter steps run");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)P.S * P.F;
  if (mean) HIPCHK(h, hipMemcpy(mean, P.mean.p, n * 2 * sizeof(double), hipMemcpyDeviceToHost));
  if (range_error) HIPCHK(h, hipMemcpy(range_error, P.err.p, n * sizeof(double), hipMemcpyDeviceToHost));
  if (list_len) HIPCHK(h, hipMemcpy(list_len, P.out_len.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_pf_status(auvp_handle* h, int32_t* status, uint64_t* n_draw32) {
  if (!h) return AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready) return fail(h, AUVP_ERR_STATE, "no particle-filter batch");
  HIPCHK(h, hipSetDevice(h->device));
  if (status) HIPCHK(h, hipMemcpy(status, P.status.p, (size_t)P.F * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (n_draw32) HIPCHK(h, hipMemcpy(n_draw32, P.ndraw.p, (size_t)P.F * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_pf_rng_state(auvp_handle* h, uint32_t* mt, int32_t* mt_pos) {
  if (!h) return AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready) return fail(h, AUVP_ERR_STATE, "no particle-filter batch");
  HIPCHK(h, hipSetDevice(h->device));
  if (mt) HIPCHK(h, hipMemcpy(mt, P.mt.p, (size_t)P.F * 624 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (mt_pos) HIPCHK(h, hipMemcpy(mt_pos, P.mtpos.p, (size_t)P.F * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_pf_step_log(auvp_handle* h, double* updated, int32_t* choice) {
  if (!h) return AUVP_ERR_ARG;
  PfState& P = *pf_of(h);
  if (!P.ready || !P.have_log) return fail(h, AUVP_ERR_STATE, "the last auvp_pf_run did not record a step log");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)P.S * P.F * P.N;
  if (updated) HIPCHK(h, hipMemcpy(updated, P.updated.p, n * 5 * sizeof(double), hipMemcpyDeviceToHost));
  if (choice) HIPCHK(h, hipMemcpy(choice, P.choice.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

}  // extern "C"
#endif
