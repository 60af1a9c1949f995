// lart_host_line.cpp -- the resonance + fluorescence multiplets of setup_resonance_line (line types 4, 5, 6; line_mod.f90:713-1153)
// for both hosts: lart_host_line_lookup(par%line_id) -> lart_line_multiplet.  CPU only.
#include "../../include/lart_gpu.h"
#include <cmath>
#include <cstring>

namespace {
struct Row { const char *id, *id2, *ion; int type; double wl, mass; int nup; double f12[3], level_cm[3]; int ndown[3];
             double A21[3][4], Elow_cm[3][4], E1[3][4], E2[3][4], E3[3][4]; };
const Row ROWS[] = {
#include "line_table.inc"
};
// random_alias_setup (random_mt.f90:2016-2067): probs scaled by n in place, 1-based alias, the two work stacks popped from the top
void alias_setup(int n, double *probs, int32_t *alias) {
    int small[LART_LINE_MAX_DOWN], large[LART_LINE_MAX_DOWN], ns = 0, nl = 0;
    for (int i = 1; i <= n; i++) { alias[i - 1] = 0; probs[i - 1] = n * probs[i - 1]; if (probs[i - 1] < 1.0) small[ns++] = i; else large[nl++] = i; }
    while (ns > 0 && nl > 0) {
        const int ks = small[--ns], kl = large[--nl];
        alias[ks - 1] = kl;
        probs[kl - 1] = probs[kl - 1] + probs[ks - 1] - 1.0;
        if (probs[kl - 1] < 1.0) small[ns++] = kl; else large[nl++] = kl;
    }
}
}  // namespace

extern "C" int lart_host_line_lookup(const char *line_id, lart_line_multiplet *m) {
    if (!line_id || !m) return -1;
    const double sigma_0 = 0.026540083434, amu = 1.67262192e-24, vtherm1_amu = 0.12895319011972164, h_planck = 6.62607004e-34, um2m = 1.0e-6;
    const double speedc_cm = 2.99792458e5 * 1.0e5, pi = 3.141592653589793238462643383279502884197;
    for (const Row &r : ROWS) {
        if (std::strcmp(line_id, r.id) != 0 && (r.id2[0] == 0 || std::strcmp(line_id, r.id2) != 0)) continue;
        std::memset(m, 0, sizeof *m);
        m->line_type = r.type; m->nup = r.nup; std::strncpy(m->ion_id, r.ion, sizeof m->ion_id - 1);
        m->wavelength0 = r.wl; m->cross0 = sigma_0 / std::sqrt(pi) * r.f12[0]; m->vtherm1 = vtherm1_amu / std::sqrt(r.mass);
        m->g_recoil0 = (h_planck / amu / r.mass) / ((r.wl * um2m) * (r.wl * um2m));
        for (int i = 0; i < r.nup; i++) {
            m->f12[i] = r.f12[i]; m->ndown[i] = r.ndown[i];
            m->delE_Hz[i] = r.nup > 1 ? r.level_cm[0] * speedc_cm - r.level_cm[i] * speedc_cm : 0.0;     // :833-834
            double sum = 0.0;
            for (int d = 0; d < r.ndown[i]; d++) { m->A21[i][d] = r.A21[i][d]; sum += r.A21[i][d]; m->Elow_Hz[i][d] = r.Elow_cm[i][d] * speedc_cm;
                                                   m->E1[i][d] = r.E1[i][d]; m->E2[i][d] = r.E2[i][d]; m->E3[i][d] = r.E3[i][d]; }
            m->damping_up[i] = sum;
            if (r.ndown[i] > 1) { for (int d = 0; d < r.ndown[i]; d++) m->P_down[i][d] = r.A21[i][d] / sum; alias_setup(r.ndown[i], m->P_down[i], m->A_down[i]); }
            else { m->P_down[i][0] = 1.0; m->A_down[i][0] = 1; }
        }
        m->damping = m->damping_up[0];
        return 0;
    }
    return -1;
}
