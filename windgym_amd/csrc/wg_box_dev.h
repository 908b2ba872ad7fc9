// wg_box_dev.h — lookups of the frozen turbulence boxes (fine box / block-averaged meandering box / isotropic wake-added box),
// shared by the flow kernels (wg_flow.hip: k_flow, k_windspeed; wg_envb.hip: k_flow_envb).  Inline per translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include "wg_flow.h"

// trilinear, periodic lookup of the frozen box at (x, y, z) metres.  The box is stored interleaved
// ([Nx][Ny][Nz] cells of float4 = (u, v, w, 0), repacked once in wg_set_turbulence_box): the two z-neighbours
// of a corner are 32 contiguous bytes and one 16-byte load brings all components, instead of 8 scattered 4-byte
// loads per component.  Cell coordinates in double precision (x - U t reaches 1e5 m), weights in fp32 — as the
// oracle does.  out[0..2] = (u, v, w).
template <bool POW2>
__device__ __forceinline__ void box_lookup_dims(const float4* __restrict__ box, const int bnx, const int bny, const int bnz,
                                                const double inv_dx, const double inv_dy, const double inv_dz, double x,
                                                double y, double z, float* __restrict__ out) {
    const double fx = x * inv_dx, fy = y * inv_dy, fz = z * inv_dz;
    const double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    const float tx = (float)(fx - ix), ty = (float)(fy - iy), tz = (float)(fz - iz);
    // |cell index| < 2^31 for any realistic episode (x - U t < 1e6 m); power-of-two boxes wrap with a mask
    int i0, j0, k0, i1, j1, k1;
    if (POW2) {
        i0 = (int)ix & (bnx - 1); j0 = (int)iy & (bny - 1); k0 = (int)iz & (bnz - 1);
        i1 = (i0 + 1) & (bnx - 1); j1 = (j0 + 1) & (bny - 1); k1 = (k0 + 1) & (bnz - 1);
    } else {
        i0 = (int)ix % bnx; if (i0 < 0) i0 += bnx;
        j0 = (int)iy % bny; if (j0 < 0) j0 += bny;
        k0 = (int)iz % bnz; if (k0 < 0) k0 += bnz;
        i1 = i0 + 1 == bnx ? 0 : i0 + 1; j1 = j0 + 1 == bny ? 0 : j0 + 1; k1 = k0 + 1 == bnz ? 0 : k0 + 1;
    }
    // cell (i, j, k) lives at X(i) + Y(j) + Z(k): in 4 x 4 x 4 BRICKS of 1 KB when every dimension is a multiple of 4
    // (wg_box_cell; the repack kernels write that order), plain [Nx][Ny][Nz] otherwise.  The 8 corners of a point then
    // fall into 2-3 cache lines instead of 4 (the z pair is contiguous either way, the y neighbour is 64 B away instead of
    // Nz x 16 B, the x neighbour 256 B instead of a whole plane): the rotor-point lookups of cfg5 — fine box and
    // wake-added box, 680 of the 1050 MB a launch moved — fetch a third fewer lines.
    const bool brick = ((bnx | bny | bnz) & 3) == 0;
    const size_t nbz = (size_t)(bnz >> 2), nbyz = (size_t)(bny >> 2) * nbz;
    const size_t X0 = brick ? (size_t)(i0 >> 2) * nbyz * 64 + (size_t)(i0 & 3) * 16 : (size_t)i0 * bny * bnz;
    const size_t X1 = brick ? (size_t)(i1 >> 2) * nbyz * 64 + (size_t)(i1 & 3) * 16 : (size_t)i1 * bny * bnz;
    const size_t Y0 = brick ? (size_t)(j0 >> 2) * nbz * 64 + (size_t)(j0 & 3) * 4 : (size_t)j0 * bnz;
    const size_t Y1 = brick ? (size_t)(j1 >> 2) * nbz * 64 + (size_t)(j1 & 3) * 4 : (size_t)j1 * bnz;
    const size_t Z0 = brick ? (size_t)(k0 >> 2) * 64 + (size_t)(k0 & 3) : (size_t)k0;
    const size_t Z1 = brick ? (size_t)(k1 >> 2) * 64 + (size_t)(k1 & 3) : (size_t)k1;
    const float4 v000 = box[X0 + Y0 + Z0], v100 = box[X1 + Y0 + Z0], v010 = box[X0 + Y1 + Z0], v110 = box[X1 + Y1 + Z0];
    const float4 v001 = box[X0 + Y0 + Z1], v101 = box[X1 + Y0 + Z1], v011 = box[X0 + Y1 + Z1], v111 = box[X1 + Y1 + Z1];
#define WG_TRI(f)                                                         \
    ([&]() {                                                              \
        const float c00 = v000.f + tx * (v100.f - v000.f);                \
        const float c10 = v010.f + tx * (v110.f - v010.f);                \
        const float c01 = v001.f + tx * (v101.f - v001.f);                \
        const float c11 = v011.f + tx * (v111.f - v011.f);                \
        const float d0 = c00 + ty * (c10 - c00);                          \
        const float d1 = c01 + ty * (c11 - c01);                          \
        return d0 + tz * (d1 - d0);                                       \
    }())
    out[0] = WG_TRI(x); out[1] = WG_TRI(y); out[2] = WG_TRI(z);
#undef WG_TRI
}

template <bool POW2>
__device__ __forceinline__ void box_lookup(const float4* __restrict__ box, const FlowP& p, double x, double y,
                                           double z, float* __restrict__ out) {
    box_lookup_dims<POW2>(box, p.bnx, p.bny, p.bnz, p.inv_bdx, p.inv_bdy, p.inv_bdz, x, y, z, out);
}
// the isotropic box of the wake-added turbulence (FlowP::added)
__device__ __forceinline__ void abox_lookup(const FlowP& p, const FlowPtrs& d, double x, double y, double z,
                                            float* __restrict__ out) {
    if (p.abox_pow2) box_lookup_dims<true>(d.abox4, p.anx, p.any, p.anz, p.inv_adx, p.inv_ady, p.inv_adz, x, y, z, out);
    else box_lookup_dims<false>(d.abox4, p.anx, p.any, p.anz, p.inv_adx, p.inv_ady, p.inv_adz, x, y, z, out);
}

// The wake particles read the transverse components from the meandering box: the field block-averaged over
// 4x4x4 cells (coarse cell i is centred at fine index 4 i + 1.5), stored [cny][cnz][cnx] with x FASTEST as cells of
// (v_k, w_k, v_k+1, w_k+1): the two z-neighbours of a corner come with ONE aligned 16-byte load (4 gathers per
// particle instead of 8), and because the particles of a chain are 0.2 D apart along x the lanes of a wave
// (consecutive ring slots) share a few cache lines per row instead of touching 64.  17 MB for the reference's
// 0.8 GB box: it lives in L2 / Infinity Cache.  (Measured alternatives: 8-byte cells with unaligned x-pair loads
// halve the footprint but the compiler splits them into 8 gathers: -6 %.)
template <bool POW2>
__device__ __forceinline__ void cbox_lookup_vw_dims(const float4* __restrict__ box, const int bnx, const int bny, const int bnz,
                                                    const double inv_bdx, const double inv_bdy, const double inv_bdz, double x,
                                                    double y, double z, float& fv, float& fw) {
    const double fx = (x * inv_bdx - 1.5) * 0.25, fy = (y * inv_bdy - 1.5) * 0.25, fz = (z * inv_bdz - 1.5) * 0.25;
    const double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    const float tx = (float)(fx - ix), ty = (float)(fy - iy), tz = (float)(fz - iz);
    int i0, j0, k0, i1, j1;
    if (POW2) {
        i0 = (int)ix & (bnx - 1); j0 = (int)iy & (bny - 1); k0 = (int)iz & (bnz - 1);
        i1 = (i0 + 1) & (bnx - 1); j1 = (j0 + 1) & (bny - 1);
    } else {
        i0 = (int)ix % bnx; if (i0 < 0) i0 += bnx;
        j0 = (int)iy % bny; if (j0 < 0) j0 += bny;
        k0 = (int)iz % bnz; if (k0 < 0) k0 += bnz;
        i1 = i0 + 1 == bnx ? 0 : i0 + 1; j1 = j0 + 1 == bny ? 0 : j0 + 1;
    }
    const size_t r0 = ((size_t)j0 * bnz + k0) * bnx, r1 = ((size_t)j1 * bnz + k0) * bnx;
    const float4 c00 = box[r0 + i0], c10 = box[r0 + i1];
    const float4 c01 = box[r1 + i0], c11 = box[r1 + i1];
    // same association as box_lookup / the oracle: x, then y, then z
#define WG_TRI2(lo, hi)                                                   \
    ([&]() {                                                              \
        const float e00 = c00.lo + tx * (c10.lo - c00.lo);                \
        const float e10 = c01.lo + tx * (c11.lo - c01.lo);                \
        const float e01 = c00.hi + tx * (c10.hi - c00.hi);                \
        const float e11 = c01.hi + tx * (c11.hi - c01.hi);                \
        const float d0 = e00 + ty * (e10 - e00);                          \
        const float d1 = e01 + ty * (e11 - e01);                          \
        return d0 + tz * (d1 - d0);                                       \
    }())
    fv = WG_TRI2(x, z); fw = WG_TRI2(y, w);
#undef WG_TRI2
}

template <bool POW2>
__device__ __forceinline__ void cbox_lookup_vw(const float4* __restrict__ box, const FlowP& p, double x, double y,
                                               double z, float& fv, float& fw) {
    cbox_lookup_vw_dims<POW2>(box, p.cnx, p.cny, p.cnz, p.inv_bdx, p.inv_bdy, p.inv_bdz, x, y, z, fv, fw);
}
