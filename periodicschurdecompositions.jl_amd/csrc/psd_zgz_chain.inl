// Window transfer wrappers, rotation records and the chain link of the complex signed engine.  Included by psd_zgz.h
// twice (global scope with PSD_NS = ::, namespace psd_wv with PSD_NS = psd_wv::); see psd_rgz_chain.inl.

PSD_D void psd_zgwin_load(const psd_zgparams& P, const psd_zwin& w, int n, int p, int j0 = 0, int jstep = 1) {
    psd_zparams Q;
    Q.H = P.H;
    PSD_NS psd_zwin_load(Q, w, n, p, j0, jstep);
}
PSD_D void psd_zgwin_store(const psd_zgparams& P, const psd_zwin& w, int n, int p, int j0 = 0, int jstep = 1) {
    psd_zparams Q;
    Q.H = P.H;
    PSD_NS psd_zwin_store(Q, w, n, p, j0, jstep);
}
PSD_D void psd_zgwin_set2(const psd_zwin& w, int l, int r1, int c1, psd_z v1, int r2, int c2, psd_z v2) {
    PSD_WAVE_SYNC();
    PSD_ONE {
        w.at(l, r1, c1) = v1;
        w.at(l, r2, c2) = v2;
    }
    PSD_WAVE_SYNC();
}
PSD_D void psd_zgrecord(const psd_zgparams& P, int* lcnt, int m, int pos, double c, psd_z s) {
    PSD_ONE {
        const int q = lcnt[m - 1];
        if (q < PSD_GTR_CAP) {
            psd_ztr tr;
            tr.pos = pos;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.tr[(size_t)(m - 1) * PSD_GTR_CAP + q] = tr;
        }
        lcnt[m - 1] = q + 1;
    }
    PSD_WAVE_SYNC();
}
// One factor of a rotation chain inside the window (see psd_g_link): incoming (c, s) at (q, q+1);
//   cols_in:  right on the columns, new row rotation from (H[q,q], H[q+1,q])          (generalized.jl:823-832)
//   !cols_in: left on the rows, new column rotation from (H[q+1,q+1], H[q+1,q]): the reference's backwards
//             Givens(q+1, q, c, s') is the standard rotation (q, q+1; c, -s)            (:833-845)
PSD_D void psd_zg_link(const psd_zwin& w, int l, int q, bool cols_in, double& c, psd_z& s, int rlo, int chi) {
    psd_z r;
    if (cols_in) {
        PSD_NS psd_zwin_right(w, l, q, c, s, rlo, q + 1);
        psd_zgivens(w.at(l, q, q), w.at(l, q + 1, q), c, s, r);
        PSD_NS psd_zgwin_set2(w, l, q, q, r, q + 1, q, zmk(0.0, 0.0));
        PSD_NS psd_zwin_left(w, l, q, c, s, q + 1, chi);
    } else {
        PSD_NS psd_zwin_left(w, l, q, c, s, q, chi);
        psd_zgivens(w.at(l, q + 1, q + 1), zneg(w.at(l, q + 1, q)), c, s, r);
        PSD_NS psd_zgwin_set2(w, l, q + 1, q + 1, r, q + 1, q, zmk(0.0, 0.0));
        PSD_NS psd_zwin_right(w, l, q, c, s, rlo, q);
    }
}

