// pf_types.h -- the launch record of the particle-filter kernels (pf_kernel.h), shared by the kernels' translation unit
// (pf_kernels.hip) and the host side (pf_host.h in auvplan.hip).
#ifndef AUVP_PF_TYPES_H
#define AUVP_PF_TYPES_H
#include <stdint.h>

namespace auvp {

enum { PF_PHASE_UPDATE = 1, PF_PHASE_WEIGHTS = 2, PF_PHASE_MEAN = 4 };
enum { PF_OK = 0, PF_ERR_ANGLE = 1, PF_ERR_EMPTY = 2 };

struct PfDev {
  int32_t F, N, A, S, phases, _pad;
  double* st;              // [F][5][N]  x, y, v, theta, weight per list position
  int32_t* ent;            // [F][N]     object id of the list position (index drawn by the last correct)
  int32_t* llen;           // [F]        bound of the object ids (len(list_of_new_particles) of the last correct)
  uint32_t* mt;            // [F][624]
  int32_t* mtpos;          // [F]
  const double* shark0;    // [F][2]     (create)
  const double* meas;      // [S][F][A][5]
  const double* shark;     // [S][F][2]
  double* mean;            // [S][F][2]
  double* err;             // [S][F]
  int32_t* out_len;        // [S][F]
  int32_t* status;         // [F]
  unsigned long long* ndraw;  // [F]
  double* updated;         // [S][F][N][5] or null (diagnostic)
  int32_t* choice;         // [S][F][N] or null (diagnostic)
};

}  // namespace auvp
#endif  // AUVP_PF_TYPES_H
