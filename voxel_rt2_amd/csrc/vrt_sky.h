// vrt_sky.h -- skybox parameterisation and the two lookups the render kernel performs.
//
// Replaces Atmos.project_sky / unproject_sky / sample_skybox / sample_skybox_transmittance
// (reference renderer/atmos.py:94-131, 428-455).  The two res x res RGB tables (354 MB at the
// reference's 3840^2) are the only data on the hot path that does not fit the Infinity Cache
// alongside the frame buffers; they are stored [u][v][3] f32 exactly like the reference fields so
// the four taps of a lookup are two 24-byte runs.
#ifndef VRT_SKY_H
#define VRT_SKY_H

#include "vrt_types.h"

namespace vrt {

VRT_DEV f2 project_sky(float fres, f3 d) {  // atmos.py:428-440
    float pl = 1.0f / dm_sqrt(d.x * d.x + d.z * d.z);
    f2 pd = mk2(pl * d.x, pl * d.z);
    const float horizon = (float)(3.141592653589793 * 0.5);
    float azimuth = DM_PI + dm_atan2(pd.x, -pd.y);
    float elevation = horizon - dm_acos(d.y);
    float cx = azimuth / DM_TWO_PI;
    float cy = 0.5f + 0.5f * sgn(elevation) * dm_sqrt((float)(2.0 / 3.141592653589793) * dm_abs(elevation));
    return mk2(cx * (1.0f - fres) + 0.5f * fres, cy * (1.0f - fres) + 0.5f * fres);
}
VRT_DEV f3 unproject_sky(float fres, f2 uv) {  // atmos.py:442-455
    float cx = (uv.x - 0.5f * fres) / (1.0f - 1.0f * fres);
    float cy = (uv.y - 0.5f * fres) / (1.0f - 1.0f * fres);
    cy = (cy < 0.5f) ? -sq(1.0f - 2.0f * cy) : sq(2.0f * cy - 1.0f);
    float azimuth = cx * 2.0f * DM_PI - DM_PI;
    float elevation = cy * 0.5f * DM_PI;
    float se, ce, sa, ca;
    dm_sincos(elevation, &se, &ce);
    dm_sincos(azimuth, &sa, &ca);
    return norm3(mk3(ce * sa, se, -ce * ca));
}

struct SkyTap { int i00, i10, i01, i11; float fx, fy; };
VRT_DEV SkyTap sky_taps(const SkyTables& t, f2 tc) {
    float fcx = tc.x * (float)t.res - 0.5f, fcy = tc.y * (float)t.res - 0.5f;
    int ix = dm_f2i(fcx), iy = dm_f2i(fcy);
    ix = ix < 0 ? 0 : (ix > t.res - 1 ? t.res - 1 : ix);   // only NaN directions can leave the table
    iy = iy < 0 ? 0 : (iy > t.res - 1 ? t.res - 1 : iy);
    int ix1 = (ix + 1) % t.res, iy1 = (iy + 1) % t.res;
    SkyTap s;
    s.i00 = ix * t.res + iy; s.i10 = ix1 * t.res + iy; s.i01 = ix * t.res + iy1; s.i11 = ix1 * t.res + iy1;
    s.fx = frac1(fcx); s.fy = frac1(fcy);
    return s;
}
VRT_DEV f3 ld3(const float* p, int i) { return mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
VRT_DEV f3 sky_bilinear(const float* tab, const SkyTap& s) {
    f3 bl = ld3(tab, s.i00), br = ld3(tab, s.i10), tl = ld3(tab, s.i01), tr = ld3(tab, s.i11);
    return lerp3(lerp3(bl, br, s.fx), lerp3(tl, tr, s.fx), s.fy);
}
// atmos.py:117-131
VRT_DEV f3 sky_transmittance(const SkyTables& t, f3 d) {
    return sky_bilinear(t.transmittance, sky_taps(t, project_sky(t.fres, d)));
}
// atmos.py:94-115 (direction jittered by three draws)
VRT_DEV void sky_lookup(const SkyTables& t, f3 d, dm_rng& rng, f3& scat, f3& trans) {
    float r0 = dm_rng_f32(&rng), r1 = dm_rng_f32(&rng), r2 = dm_rng_f32(&rng);
    SkyTap s = sky_taps(t, project_sky(t.fres, norm3(d + mk3(r0, r1, r2) * 0.0015f)));
    scat = sky_bilinear(t.scattering, s);
    trans = sky_bilinear(t.transmittance, s);
}

}  // namespace vrt
#endif
