// sightline.hpp -- sight-line tau(nu) / N(HI) maps per observer pixel (par%save_sightline_tau).
// Reference: make_sightline_tau_{outside,outside_amr} (sightline_tau_rect.f90:11-190, :192-345),
// make_sightline_tau_{inside,inside_amr} (sightline_tau_heal.f90:11-165, :167-290), make_sightline_tau_clump
// (sightline_tau_clump.f90:33-155).  The host lays one ray per pixel down (start point on the far boundary, direction
// towards the observer); the kernel marches every ray once per frequency bin.
#pragma once
#include "lart_device.hpp"
#include <vector>

// one sight line.  dist < 0: march to the edge of the medium; >= 0: stop after that length (observer inside the medium).
// ic/jc/kc: start cell of the Cartesian grid; octree: ic = 0, the kernel looks the leaf up; clumps: unused (start in vacuum).
struct SightRay { double x, y, z, kx, ky, kz, dist; int ic, jc, kc, ok; };

// what the ray set-up needs from the context, host side
struct SightGeom {
    int var;                                   // 0 Cartesian, 3 octree, 4 clumps
    int nside, nobs, nxim, nyim; double dxim, dyim;
    const DObs *obs;                           // host copy
    int nx, ny, nz; double xmin, ymin, zmin, dx, dy, dz;
    const double *xface, *yface, *zface;       // host copies, n+1 each (Cartesian)
    double lo, hi;                             // octree box
    double sphere_R;                           // clump medium
};

void lart_sight_pix2vec(int nside, int ipix, double v[3]);
void lart_sight_build_rays(const SightGeom &g, std::vector<SightRay> &rays);
// launches on `stream`; tau_gas[nray*nxfreq], N_gas[nray], tau_dust[nray] are device pointers (tau_dust may be null)
hipError_t lart_sight_launch(const KArgs &A, int var, const SightRay *d_rays, int nray, double cross0, bool dust,
                             double *d_tau_gas, double *d_N_gas, double *d_tau_dust, size_t shmem, hipStream_t stream);
