// compose_host.h -- entry points that chain two stages of the path on the device (included after the planner and
// particle-filter hosts: same translation unit).
#ifndef AUVP_COMPOSE_HOST_H
#define AUVP_COMPOSE_HOST_H

extern "C" {

// Config 5 (BASELINE.json configs[4]; particleFilter.py:283-317 feeding gym_rrt/envs/rrt_dubins.py:205-248): every
// particle of the handle's filter batch becomes the goal of one Planner_RRT episode.  Everything stays on the device:
// the goals are read out of the filters' particle state, the generator of episode e is seeded as
// random.seed(seed_base + e) by one thread per episode, the trees are re-planted.  Then auvp_prrt_plan().
int auvp_prrt_replan_particles(auvp_handle* h, const double* start4, const auvp_prrt_params* p, const double* xform,
                               const double* clamp4, uint64_t seed_base, int32_t flags) {
  if (!h || !start4 || !p || !xform || !clamp4) return h ? fail(h, AUVP_ERR_ARG, "null argument") : AUVP_ERR_ARG;
  PfState& F = *pf_of(h);
  if (!F.ready) return fail(h, AUVP_ERR_STATE, "no particle-filter batch on this handle (auvp_pf_create_batch)");
  const long long EE = (long long)F.F * F.N;
  if (EE > 0x7fffffffLL) return fail(h, AUVP_ERR_ARG, "too many particles");
  const int32_t E = (int32_t)EE;
  HIPCHK(h, hipSetDevice(h->device));
  PrrtState& S = *prrt_of(h);
  int rc;
  if ((rc = prrt_configure(h, S, E, p, flags))) return rc;
  auvp::PrrtGoalMap M;
  if ((rc = upload(h, h->d_tmp4, xform, (size_t)F.F * 4))) return rc;
  for (int i = 0; i < 4; i++) { M.start[i] = start4[i]; M.clamp[i] = clamp4[i]; }
  M.xform = h->d_tmp4.as<double>();
  M.seed_base = seed_base; M.n_filters = F.F; M.n_particles = F.N;
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(auvp::prrt_from_particles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                auvp::PRRT_SEED_LDS));
  hipLaunchKernelGGL(auvp::prrt_from_particles_kernel, dim3((E + 63) / 64), dim3(64), auvp::PRRT_SEED_LDS, h->stream, S.B, M,
                     F.st.as<double>(), (int)E);
  HIPCHK(h, hipGetLastError());
  rc = prrt_plant(h, S, E);
  if (rc != AUVP_OK) return rc;
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  return AUVP_OK;
}

}  // extern "C"
#endif
