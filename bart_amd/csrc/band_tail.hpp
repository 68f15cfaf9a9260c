// Band integration and energy balance as the tail of the RT kernel (round 5).
//
// The per-step path ends with wine.bandintegrate (code/wine.py:177-199; BARTfunc.py:386-396) and the optional energy
// check (BARTfunc.py:366-383) on the spectrum the RT kernel has just written: `step_bandflux`, a launch of its own,
// 9 us + a launch boundary at the headline shape.  Both are dot products of the spectrum with fixed weights -- the
// trapezoid rule written per sample, sum_j F_j q_j with q_j = (filter weight)_j x (half the distance between j's
// neighbours inside the window) -- so every workgroup of the single-wave kernel can add its 64 samples' share while
// it still holds them: a wave-wide sum per filter whose window meets the tile (DPP row scans, no LDS), stored to
// part[walker][tile][F + 1]; the LAST workgroup of a walker to deliver (a counter per walker) adds the tiles' shares in
// tile order -- the same order every run: band fluxes are reproducible to the bit -- applies the status rules and
// writes band[walker][F].  Stores and loads of the shares are agent-scope atomics (write-through / L2 bypass: the
// workgroups of a walker sit on different XCDs whose L2s are not coherent with each other), the counter an agent-scope
// atomic add issued after the stores have been acknowledged.
#pragma once
#include "kernels.hpp"

namespace bartrt {

template <int CTRL>
__device__ __forceinline__ double band_dpp(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// sum over the 64 lanes in a fixed order, the same value in every lane: inclusive scans inside the four 16-lane rows
// (row_shr 1, 2, 4, 8 with zero fill), then the rows' totals (lanes 15, 31, 47, 63) added as (r0 + r1) + (r2 + r3)
__device__ __forceinline__ double band_wave_sum(double v) {
  v += band_dpp<0x111>(v);
  v += band_dpp<0x112>(v);
  v += band_dpp<0x114>(v);
  v += band_dpp<0x118>(v);
  auto row = [&](int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
  };
  return (row(15) + row(31)) + (row(47) + row(63));
}

__device__ __forceinline__ void band_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double band_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One wave = one tile of 64 wavenumbers of walker w; lane = wavenumber i, F = its flux (valid: i < W).
__device__ __forceinline__ void band_tail(const RtArgs &p, int w, int tile, int i, bool valid, double F) {
  const BandDev &b = *p.band;
  const int nF = b.F, lane = threadIdx.x & 63;
  double *part = b.part + ((size_t)w * p.ntiles + tile) * (size_t)(nF + 1);
  const int t0 = tile * 64, t1 = t0 + 64;
  for (int f = 0; f < nF; f++) {
    const int i0 = b.idx0[f], n = b.npts[f];
    double s = 0.0;
    if (i0 < t1 && i0 + n > t0) {          // (wave-uniform: the filter's window meets this tile)
      const int j = i - i0;
      const bool in = valid && j >= 0 && j < n;
      s = band_wave_sum(in ? F * b.q[b.woff[f] + j] : 0.0);
    }
    if (lane == 0) band_store(part + f, s);
  }
  {
    const double s = b.ebalance ? band_wave_sum(valid ? F * b.qe[i] : 0.0) : 0.0;
    if (lane == 0) band_store(part + nF, s);
  }
  unsigned delivered = 0;
  if (lane == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the shares are at the coherence point before the counter moves
    delivered = __hip_atomic_fetch_add(b.count + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  delivered = (unsigned)__builtin_amdgcn_readfirstlane((int)delivered);
  if (delivered != (unsigned)p.ntiles - 1u) return;
  // ---- the walker's last workgroup: its band fluxes
  if (lane == 0) __hip_atomic_store(b.count + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  const int st = p.band_status[w];
  double *out = p.band_out + (size_t)w * nF;
  if (st != 0 && st != 3) {                // rejected before the engine ran (BARTfunc.py:327-344): -1 in every band
    if (lane < nF) out[lane] = -1.0;
    for (int f = 64 + lane; f < nF; f += 64) out[f] = -1.0;
    if (p.band_status_out && lane == 0) p.band_status_out[w] = st;
    return;
  }
  const double *all = b.part + (size_t)w * p.ntiles * (size_t)(nF + 1);
  auto total = [&](int f) {
    double acc = 0.0;
    for (int t = lane; t < p.ntiles; t += 64) acc += band_load(all + (size_t)t * (nF + 1) + f);
    return band_wave_sum(acc);
  };
  if (b.ebalance && total(nF) * b.e_fac > b.e_in) {   // BARTfunc.py:377-383
    if (lane < nF) out[lane] = -1.0;
    for (int f = 64 + lane; f < nF; f += 64) out[f] = -1.0;
    if (lane == 0) {
      p.band_status[w] = 3;
      if (p.band_status_out) p.band_status_out[w] = 3;
    }
    return;
  }
  for (int f = 0; f < nF; f++) {
    const double s = total(f);
    if (lane == 0) out[f] = s;
  }
  if (p.band_status_out && lane == 0) p.band_status_out[w] = 0;
}

}  // namespace bartrt
