// velocity.h — velocity-profile generator (reference src/velocity.py:14-76, src/vehicle.py:24-35,
// src/vehicleMX5.py:19-38): one thread per profile.  The two passes are sequential scans (each sample's limit depends on the
// previous sample's final value), so the parallelism is over profiles; the rolled / flipped index arithmetic of the
// reference (start at the slowest point, np.roll / np.flip) is done on the fly on the original indices.
#pragma once
#include "layout.h"

namespace ltompc {

// (every function below: `#pragma clang fp contract(off)`: the reference is plain IEEE double arithmetic (numpy / math), so
//  no fused multiply-adds here; results then agree with it to the last bit where sqrt / division are correctly rounded)
__device__ __forceinline__ double vp_engine_force(const ltompc_vp_vehicle& V, const double v) {
#pragma clang fp contract(off)
  if (V.kind == 1) return (V.T * V.C_m) - V.Cr_0 - (V.Cr_2 * (v * v));  // vehicleMX5.py:19-21
  // vehicle.py:24-26: np.interp(velocity, map_v, map_f) (clamped at both ends)
  const int n = V.n_map;
  if (v <= V.map_v[0]) return V.map_f[0];
  if (v >= V.map_v[n - 1]) return V.map_f[n - 1];
  int j = 0;
  while (j + 2 < n && v >= V.map_v[j + 1]) j++;
  const double slope = (V.map_f[j + 1] - V.map_f[j]) / (V.map_v[j + 1] - V.map_v[j]);
  return slope * (v - V.map_v[j]) + V.map_f[j];
}
__device__ __forceinline__ double vp_traction(const ltompc_vp_vehicle& V, const double v, const double k) {
#pragma clang fp contract(off)
  const double grav = 9.81;
  double f, f_lat;
  if (V.kind == 1) {  // vehicleMX5.py:23-38
    const double Fn = V.mass * grav;
    f = V.lam * V.D * Fn;
    f_lat = V.mass * v * v * k;
  } else {  // vehicle.py:28-35
    f = V.friction_coef * V.mass * grav;
    f_lat = V.mass * (v * v) * k;
  }
  if (f <= f_lat) return 0.0;
  return sqrt(f * f - f_lat * f_lat);
}

__global__ void k_velocity_profile(ltompc_vp_vehicle V, int n, int batch, const double* __restrict__ s_all, const double* __restrict__ k_all,
                                   const double* __restrict__ s_max_all, double* __restrict__ v_all, double* __restrict__ vloc_all,
                                   double* __restrict__ vacc_all, double* __restrict__ vdec_all) {
#pragma clang fp contract(off)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double* s = s_all + (size_t)b * n;
  const double* kk = k_all + (size_t)b * n;
  double* vloc = vloc_all + (size_t)b * n;
  double* va = vacc_all + (size_t)b * n;
  double* vd = vdec_all + (size_t)b * n;
  const double s_max = s_max_all[b];
  const bool closed = s_max >= 0.0;
  const double grav = 9.81;
  // velocity.py:28-29 limit_local_velocities, and the slowest point (np.argmin: first minimum)
  int m = 0;
  double vmin = 0.0;
  for (int i = 0; i < n; i++) {
    const double vl = sqrt(V.friction_coef * grav / kk[i]);
    vloc[i] = vl, va[i] = vl, vd[i] = vl;
    if (i == 0 || vl < vmin) vmin = vl, m = i;
  }
  // velocity.py:31-53 limit_acceleration: rolled index i <-> original (i + m) % n
  const int wrap_a = (n - m) % n;
  for (int i = 0; i < n; i++) {
    if (i == wrap_a && !closed) continue;
    const int cur = (i + m) % n, prev = (i - 1 + m + n) % n;
    const double vp = va[prev];
    if (va[cur] > vp) {
      const double traction = vp_traction(V, vp, kk[prev]);
      const double eng = vp_engine_force(V, vp);
      const double force = eng < traction ? eng : traction;
      const double accel = force / V.mass;
      const double ds = i == wrap_a ? s_max - s[prev] : s[cur] - s[prev];
      const double vlim = sqrt(vp * vp + 2.0 * accel * ds);
      va[cur] = va[cur] < vlim ? va[cur] : vlim;
    }
  }
  // velocity.py:55-76 limit_deceleration: flipped rolled index i <-> original (n - 1 - i + m) % n
  for (int i = 0; i < n; i++) {
    if (i == m && !closed) continue;
    const int cur = (n - 1 - i + m) % n, prev = (n - i + m) % n;
    const double vp = vd[prev];
    if (vd[cur] > vp) {
      const double decel = vp_traction(V, vp, kk[prev]) / V.mass;
      const double ds = i == m ? s_max - s[cur] : s[prev] - s[cur];
      const double vlim = sqrt(vp * vp + 2.0 * decel * ds);
      vd[cur] = vd[cur] < vlim ? vd[cur] : vlim;
    }
  }
  for (int i = 0; i < n; i++) v_all[(size_t)b * n + i] = va[i] < vd[i] ? va[i] : vd[i];  // velocity.py:26
}

}  // namespace ltompc
