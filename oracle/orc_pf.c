/* orc_pf.c -- CPU restatement of the shark particle filter.
 *
 * TEST INFRASTRUCTURE (oracle/): checker only (see orc_api.h).
 *
 * Follows particleFilter.py (paths relative to /root/reference):
 *   angle_wrap / velocity_wrap                     :18-41   (recursive; one rounded add per level)
 *   Particle.__init__                              :44-53
 *   Particle.update_particle                       :55-75
 *   Particle.calc_particle_alpha / _range          :78-90
 *   Particle.weight                                :92-116
 *   ParticleFilter.normalize                       :127-151
 *   ParticleFilter.particleMean / meanError        :153-177
 *   ParticleFilter.correct                         :179-252
 *   ParticleFilter.create_and_update               :277-282
 *   ParticleFilter.update_weights                  :285-310
 *   ParticleFilter.create                          :311-317
 * driven in the order of robotSim.py:665-701.
 *
 * `random` in that module is numpy.random (particleFilter.py:8 shadows the stdlib import), i.e. the
 * global legacy RandomState: MT19937, random_sample = (a>>5, b>>6)/2^53, uniform(lo, hi) =
 * lo + (hi - lo) * random_sample, choice(n) = randint(0, n) = masked rejection on 32-bit outputs
 * (numpy/random/src/distributions/distributions.c buffered_bounded_masked_uint32; no draw when n == 1).
 * numpy is a dependency outside /root/reference; the stream is pinned by tests/golden/g10_*.npz, which
 * were captured by running the reference against the numpy of this image.
 *
 * Object identity matters: `correct` deep-copies, then draws list indices with replacement, so one
 * Particle object can sit at several list positions; create_and_update then moves it once per position.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "cpyrandom.h"
#include "orc_api.h"
#include "orc_math.h"

typedef struct { double x, y, v, th, w; } part;

static double np_double(cpy_rng* r) {
  uint32_t a = cpy_genrand32(r) >> 5, b = cpy_genrand32(r) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
static double np_uniform(cpy_rng* r, double lo, double hi) {
  const double range = hi - lo;
  return lo + range * np_double(r);
}
/* legacy RandomState.randint(0, n) for 1 <= n <= 2^32 */
static int64_t np_choice(cpy_rng* r, uint32_t n) {
  const uint32_t rng = n - 1;
  if (rng == 0) return 0;
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v;
  while ((v = (cpy_genrand32(r) & mask)) > rng) {}
  return v;
}

static int angle_wrap(double* a) {
  double ang = *a;
  for (int depth = 0; depth < 900; depth++) { /* CPython's recursion limit ends deeper chains with RecursionError */
    if (-M_PI <= ang && ang <= M_PI) { *a = ang; return 0; }
    else if (ang > M_PI) ang += (-2 * M_PI);
    else if (ang < -M_PI) ang += (2 * M_PI);
    else return -1; /* nan: the reference returns None */
  }
  return -1;
}
static double velocity_wrap(double v) {
  for (int depth = 0; depth < 900 && v > 5; depth++) v += -5;
  return v;
}

static int update_particle(part* p, cpy_rng* r, double dt) {
  p->v += np_uniform(r, 0, 5);
  p->v = velocity_wrap(p->v);
  p->th += np_uniform(r, -(M_PI / 2), M_PI / 2);
  if (angle_wrap(&p->th)) return -1;
  p->x += p->v * ORC_COS(p->th) * dt;
  p->y += p->v * ORC_SIN(p->th) * dt;
  return 0;
}

static int weight(part* p, const double* m) {
  /* update_weights :293-295: alpha(x, y, theta); range(x, y); weight(m[3], alpha, m[4], range) */
  double pa = ORC_ATAN2((-m[1] + p->y), (p->x + -m[0])) - m[2];
  if (angle_wrap(&pa)) return -1;
  const double pr = ORC_SQRT(ORC_POW2(m[1] - p->y) + ORC_POW2(m[0] - p->x));
  const double auv_alpha = m[3], auv_range = m[4];
  const double constant = 1.2533141375;
  double d = pa - auv_alpha;
  if (angle_wrap(&d)) return -1;
  const double function_alpha = .001 + (1 / (constant) * (ORC_POW_E((-(ORC_POW2(d))) / (0.5))));
  const double function_weight = .001 + (1 / (100 * constant) * (ORC_POW_E((-(ORC_POW2(pr - auv_range))) / (20000))));
  p->w = function_weight * function_alpha;
  return 0;
}

int orc_pf_run(const orc_pf_in* in, orc_pf_out* o) {
  const int N = in->n_particles, A = in->n_auv;
  cpy_rng r;
  memcpy(r.mt, in->mt, sizeof r.mt);
  r.idx = *in->mt_pos;
  r.n_draw32 = 0;
  int status = ORC_OK;
  part* objs = (part*)malloc(sizeof(part) * (size_t)(5 * N + 1));
  part* list = (part*)malloc(sizeof(part) * (size_t)(5 * N + 1));
  int* ent = (int*)malloc(sizeof(int) * (size_t)N);
  double* wl = (double*)malloc(sizeof(double) * (size_t)N * (A > 0 ? A : 1));
  double* norm = (double*)malloc(sizeof(double) * (size_t)N);
  int* first = (int*)malloc(sizeof(int) * (size_t)(5 * N + 1));
  if (in->do_create) {
    for (int i = 0; i < N; i++) {
      part* p = &objs[i];
      p->x = in->shark0[0] + np_uniform(&r, -150, 150);
      p->y = in->shark0[1] + np_uniform(&r, -150, 150);
      p->v = np_uniform(&r, 0, 5);
      p->th = np_uniform(&r, -M_PI, M_PI);
      p->w = 1.0 / 1000; /* NUMBER_OF_PARTICLES is the literal 1000 in Particle.__init__ */
      ent[i] = i;
      if (o->created) memcpy(o->created + 5 * (size_t)i, p, sizeof(part));
    }
  } else {
    for (int i = 0; i < N; i++) {
      ent[i] = in->init_obj ? in->init_obj[i] : i;
      memcpy(&objs[ent[i]], in->init + 5 * (size_t)i, sizeof(part));
    }
  }
  for (int s = 0; s < in->n_steps && status == ORC_OK; s++) {
    /* create_and_update: one update per list position */
    for (int p = 0; p < N; p++) if (update_particle(&objs[ent[p]], &r, .1)) status = ORC_ERR_ARG;
    if (o->updated) for (int p = 0; p < N; p++) memcpy(o->updated + ((size_t)s * N + p) * 5, &objs[ent[p]], sizeof(part));
    /* update_weights */
    for (int a = 0; a < A; a++) {
      const double* m = in->meas + ((size_t)s * A + a) * 5;
      for (int p = 0; p < N; p++) if (weight(&objs[ent[p]], m)) status = ORC_ERR_ARG;
      for (int p = 0; p < N; p++) wl[(size_t)a * N + p] = objs[ent[p]].w;
    }
    /* normalize */
    for (int a = 0; a < A; a++) {
      double den = wl[(size_t)a * N];
      for (int p = 1; p < N; p++) if (wl[(size_t)a * N + p] > den) den = wl[(size_t)a * N + p];
      for (int p = 0; p < N; p++) wl[(size_t)a * N + p] = (1 / den) * wl[(size_t)a * N + p];
    }
    for (int p = 0; p < N; p++) {
      double nw = 0;
      for (int a = 0; a < A; a++) nw += wl[(size_t)a * N + p];
      norm[p] = nw;
    }
    double fden = norm[0];
    for (int p = 1; p < N; p++) if (norm[p] > fden) fden = norm[p];
    for (int p = 0; p < N; p++) norm[p] = (1 / fden) * norm[p];
    for (int p = 0; p < N; p++) objs[ent[p]].w = norm[p];
    /* correct */
    int L = 0;
    for (int p = 0; p < N; p++) {
      const part* q = &objs[ent[p]];
      const double w = q->w;
      int k = 0;
      if (w < 0.2) k = 1; else if (w < 0.4) k = 2; else if (w < 0.6) k = 3; else if (w < .8) k = 4; else if (w <= 1.0) k = 5;
      for (int c = 0; c < k; c++) list[L++] = *q;
    }
    if (o->list_len) o->list_len[s] = L;
    if (L == 0) { status = ORC_ERR_ARG; break; } /* numpy: ValueError("a must be greater than 0") */
    for (int i = 0; i < L; i++) first[i] = -1;
    for (int n = 0; n < N; n++) {
      const int x = (int)np_choice(&r, (uint32_t)L);
      ent[n] = x;
      if (first[x] < 0) first[x] = n;
      if (o->choice) o->choice[(size_t)s * N + n] = x;
      if (o->alias_first) o->alias_first[(size_t)s * N + n] = first[x];
    }
    { part* t = objs; objs = list; list = t; }
    if (o->resampled) for (int p = 0; p < N; p++) memcpy(o->resampled + ((size_t)s * N + p) * 5, &objs[ent[p]], sizeof(part));
    /* particleMean, meanError */
    double sum_x = 0, sum_y = 0;
    int count = 0;
    for (int p = 0; p < N; p++) { sum_x += objs[ent[p]].x; sum_y += objs[ent[p]].y; count += 1; }
    const double xm = sum_x / count, ym = sum_y / count;
    if (o->mean) { o->mean[2 * s] = xm; o->mean[2 * s + 1] = ym; }
    const double xd = xm - in->shark_xy[2 * s], yd = ym - in->shark_xy[2 * s + 1];
    if (o->range_error) o->range_error[s] = ORC_SQRT((ORC_POW2(xd)) + (ORC_POW2(yd)));
  }
  memcpy(in->mt, r.mt, sizeof r.mt);
  *in->mt_pos = r.idx;
  o->n_draw32 = r.n_draw32;
  free(objs); free(list); free(ent); free(wl); free(norm); free(first);
  return status;
}

/* numpy known answers: np.random.seed(seed); n x uniform(-150, 150); n x choice(choice_n) */
void orc_np_kat(uint32_t seed, int n, double* out_uniform, int32_t* out_choice, int32_t choice_n) {
  cpy_rng r;
  cpy_init_genrand(&r, seed);
  r.n_draw32 = 0;
  for (int i = 0; i < n; i++) out_uniform[i] = np_uniform(&r, -150, 150);
  for (int i = 0; i < n; i++) out_choice[i] = (int32_t)np_choice(&r, (uint32_t)choice_n);
}

double orc_pow_e(double z) { return ORC_POW_E(z); }
