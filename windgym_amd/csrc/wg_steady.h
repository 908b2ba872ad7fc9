// wg_steady.h — parameter block of k_steady (wg_steady.hip), filled by wg_steady_power (wg_api.hip).
#pragma once
struct SteadyP {
    int n_cases, N, S, n_tab, model, n_quad;
    float D, hub, R;
    float ka, kb, eps0, hill, tia, tib, tic, tid;
    double cx0, cy0;
    const double *x_pos, *y_pos;
    const float *rotor_dy, *rotor_dz, *tab_ws, *tab_power, *tab_ct;
};
