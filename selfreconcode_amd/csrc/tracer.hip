// Per-ray step of the masked Newton ray/surface refiner (SURVEY.md 8(a) row a12,
// utils/FindSurfacePs.py:114-163).  The network evaluations (SDF value + gradient, deformer value +
// Jacobian) come from the group-4 MLP kernels and the fused LBS kernel; this kernel does everything
// in between in one launch: convergence test (|f| < dthr and asin(|(d-c) x v| / |d-c|) in degrees
// < athr), residual L = w1 |f| + w2 |(d-c) x v| / |d-c|, its gradient through J_d^T, and the step
// p <- p - L g / |g|^2 -- what the reference spreads over ~25 elementwise launches, two autograd
// passes and a per-frame host-synchronising loop.
#include "sr_common.h"

namespace {
__global__ __launch_bounds__(256) void newton_kernel(sr_newton_args g) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.M; i += (int64_t)gridDim.x * blockDim.x) {
    const float f = g.sdf4[(i * g.group) * g.ld_sdf];
    const float* off = g.off4 + (i * g.group) * g.ld_off;
    const float yx = g.y[i * 3], yy = g.y[i * 3 + 1], yz = g.y[i * 3 + 2];
    const float vx = g.rays[i * 3], vy = g.rays[i * 3 + 1], vz = g.rays[i * 3 + 2];
    const float dx = yx - g.cam[0], dy = yy - g.cam[1], dz = yz - g.cam[2];
    const float ux = dy * vz - dz * vy, uy = dz * vx - dx * vz, uz = dx * vy - dy * vx;   // (d - c) x v
    const float un = sqrtf(ux * ux + uy * uy + uz * uz), dn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float s = un / dn;
    const bool ok = (fabsf(f) < g.dthreshold) && (asinf(s) * 180.0f / 3.14159265358979323846f < g.athreshold);
    g.converged[i] = ok ? 1 : 0;
    if (g.group != 4 || ok || !g.p_out) {
      if (g.p_out) { g.p_out[i * 3] = g.p[i * 3]; g.p_out[i * 3 + 1] = g.p[i * 3 + 1]; g.p_out[i * 3 + 2] = g.p[i * 3 + 2]; }
      continue;
    }
    // d s / d d = (v x u^) / |d| - |u| d / |d|^3
    const float iu = un > 0.f ? 1.f / un : 0.f;
    const float hx = ux * iu, hy = uy * iu, hz = uz * iu;
    float ex = (vy * hz - vz * hy) / dn - un * dx / (dn * dn * dn);
    float ey = (vz * hx - vx * hz) / dn - un * dy / (dn * dn * dn);
    float ez = (vx * hy - vy * hx) / dn - un * dz / (dn * dn * dn);
    // J_d = J_lbs (I + d off / d p): rows of the tangent block are d off / d p_t
    const float* jl = g.jlbs + i * 9;
    float jq[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) jq[r * 3 + c] = (r == c ? 1.f : 0.f) + off[(1 + c) * g.ld_off + r];
    // t = J_lbs^T e ; gd = J_q^T t
    const float tx = jl[0] * ex + jl[3] * ey + jl[6] * ez, ty = jl[1] * ex + jl[4] * ey + jl[7] * ez, tz = jl[2] * ex + jl[5] * ey + jl[8] * ez;
    const float gdx = jq[0] * tx + jq[3] * ty + jq[6] * tz, gdy = jq[1] * tx + jq[4] * ty + jq[7] * tz, gdz = jq[2] * tx + jq[5] * ty + jq[8] * tz;
    const float sg = f > 0.f ? 1.f : (f < 0.f ? -1.f : 0.f);
    const float* gf = g.sdf4 + (i * 4 + 1) * g.ld_sdf;
    const float gx = g.w1 * sg * gf[0] + g.w2 * gdx, gy = g.w1 * sg * gf[g.ld_sdf] + g.w2 * gdy, gz = g.w1 * sg * gf[2 * g.ld_sdf] + g.w2 * gdz;
    const float L = g.w1 * fabsf(f) + g.w2 * s;
    const float t = -L / (gx * gx + gy * gy + gz * gz);
    g.p_out[i * 3] = g.p[i * 3] + t * gx;
    g.p_out[i * 3 + 1] = g.p[i * 3 + 1] + t * gy;
    g.p_out[i * 3 + 2] = g.p[i * 3 + 2] + t * gz;
  }
}
}  // namespace

extern "C" int sr_newton_update(const sr_newton_args* a, void* stream) {
  if (!a || a->M < 0 || (a->group != 1 && a->group != 4)) return SR_EINVAL;
  if (a->M == 0) return SR_OK;
  if (!a->sdf4 || !a->y || !a->rays || !a->cam || !a->converged || !a->p) return SR_EINVAL;
  if (a->group == 4 && a->p_out && (!a->off4 || !a->jlbs)) return SR_EINVAL;
  hipLaunchKernelGGL(newton_kernel, dim3(sr_stream_grid(a->M, 256)), dim3(256), 0, (hipStream_t)stream, *a);
  return sr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Reverse-mode variant of the refiner step: the SDF gradient and the deformer vector-Jacobian product come from
// one reverse sweep each over group-1 activations (2x the rows of a value pass) instead of a group-4 forward (4x).
//   prepare: convergence test + cotangent t = J_lbs^T (d s / d y) of the deformation offset, written as rows [M, ld_t]
//   apply  : g = w1 sign(f) grad_f + w2 (t + J_off^T t),  p <- p - L g / |g|^2
namespace {
__global__ __launch_bounds__(256) void newton_prepare_kernel(sr_newton2_args g) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.M; i += (int64_t)gridDim.x * blockDim.x) {
    const float f = g.sdf[i * g.ld_sdf];
    const float vx = g.rays[i * 3], vy = g.rays[i * 3 + 1], vz = g.rays[i * 3 + 2];
    const float dx = g.y[i * 3] - g.cam[0], dy = g.y[i * 3 + 1] - g.cam[1], dz = g.y[i * 3 + 2] - g.cam[2];
    const float ux = dy * vz - dz * vy, uy = dz * vx - dx * vz, uz = dx * vy - dy * vx;
    const float un = sqrtf(ux * ux + uy * uy + uz * uz), dn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float s = un / dn;
    g.converged[i] = ((fabsf(f) < g.dthreshold) && (asinf(s) * 180.0f / 3.14159265358979323846f < g.athreshold)) ? 1 : 0;
    if (!g.t_out) continue;
    const float iu = un > 0.f ? 1.f / un : 0.f;
    const float hx = ux * iu, hy = uy * iu, hz = uz * iu;
    const float ex = (vy * hz - vz * hy) / dn - un * dx / (dn * dn * dn);
    const float ey = (vz * hx - vx * hz) / dn - un * dy / (dn * dn * dn);
    const float ez = (vx * hy - vy * hx) / dn - un * dz / (dn * dn * dn);
    const float* jl = g.jlbs + i * 9;
    float* t = g.t_out + i * g.ld_t;
    t[0] = jl[0] * ex + jl[3] * ey + jl[6] * ez;
    t[1] = jl[1] * ex + jl[4] * ey + jl[7] * ez;
    t[2] = jl[2] * ex + jl[5] * ey + jl[8] * ez;
    for (int c = 3; c < g.ld_t; ++c) t[c] = 0.f;
    g.s_out[i] = s;
  }
}

__global__ __launch_bounds__(256) void newton_apply_kernel(sr_newton2_args g) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.M; i += (int64_t)gridDim.x * blockDim.x) {
    if (g.converged[i]) { g.p_out[i * 3] = g.p[i * 3]; g.p_out[i * 3 + 1] = g.p[i * 3 + 1]; g.p_out[i * 3 + 2] = g.p[i * 3 + 2]; continue; }
    const float f = g.sdf[i * g.ld_sdf];
    const float sg = f > 0.f ? 1.f : (f < 0.f ? -1.f : 0.f);
    const float* t = g.t_out + i * g.ld_t;
    float gv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gv[c] = g.w1 * sg * g.grad_f[i * 3 + c] + g.w2 * (t[c] + g.grad_off[i * 3 + c]);
    const float L = g.w1 * fabsf(f) + g.w2 * g.s_out[i];
    const float st = -L / (gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) g.p_out[i * 3 + c] = g.p[i * 3 + c] + st * gv[c];
  }
}
}  // namespace

extern "C" int sr_newton_prepare(const sr_newton2_args* a, void* stream) {
  if (!a || a->M < 0) return SR_EINVAL;
  if (a->M == 0) return SR_OK;
  if (!a->sdf || !a->y || !a->rays || !a->cam || !a->converged) return SR_EINVAL;
  if (a->t_out && (!a->jlbs || !a->s_out || a->ld_t < 3)) return SR_EINVAL;
  hipLaunchKernelGGL(newton_prepare_kernel, dim3(sr_stream_grid(a->M, 256)), dim3(256), 0, (hipStream_t)stream, *a);
  return sr_launch_status();
}
extern "C" int sr_newton_apply(const sr_newton2_args* a, void* stream) {
  if (!a || a->M < 0) return SR_EINVAL;
  if (a->M == 0) return SR_OK;
  if (!a->sdf || !a->converged || !a->t_out || !a->s_out || !a->grad_f || !a->grad_off || !a->p || !a->p_out) return SR_EINVAL;
  hipLaunchKernelGGL(newton_apply_kernel, dim3(sr_stream_grid(a->M, 256)), dim3(256), 0, (hipStream_t)stream, *a);
  return sr_launch_status();
}
