// lrm_point_xtab.h -- BIT-EXACT reach + distance guided by the plane table (round 4).
//
// LRM_MODE_FAST used to spend ~1040 VALU instructions per point: every DECISION of distance_global
// (one_leg_global.cu:74-101 -> distance_circles one_leg.cu:321-341 -> finish_finding_closest :215-278 ->
// multi_circle_clamp :91-145) was filtered per point -- four circles, three arc tests each, ten corner points, a
// three-deep sort -- for BOTH yaw candidates, around the strict arithmetic of the winner.  The plane table of the
// tolerance mode (lrm_toltab.cpp) already holds those decisions per cell of the meridian plane: at most two clamp
// targets, at most one open point validity, and a lower bound of the in-plane distance that orders the two yaw
// candidates.  This file takes the DECISIONS from the table (with the tolerance mode's bands: a point with any
// decision inside its band is reported in `doubt` and re-evaluated by the filtered code of lrm_point_fast.h) and
// computes the VALUES with the reference's own operations in the reference's own order:
//
//   qtInvRotate, z rotation, place_over_coxa          one_leg_global.cu:76-95, one_leg.cu:9-24      strict
//   yaw = atan2f(y, x), the flipped yaw                one_leg.cu:326-328                            exact (lrm_exact_math.h)
//   per evaluated candidate: sincosf(-sat), rotation   one_leg.cu:146-156, :222-236                  exact; a candidate clamped
//                                                       to a yaw limit uses the host's sincosf(-limit) (the same floats)
//   ONE force_clamp_on_circle onto the winner          one_leg.cu:42-63                              IEEE sqrt / div
//   the yaw-limit alternative                          one_leg.cu:258-274                            filtered by the limit plane's
//                                                       offset; strict (second sincosf, two norms) when taken or near
//   restore rotation, the pick by strict norms         one_leg.cu:158-165, :334                      strict
//   place_over_coxa<Reverse>, z un-rotation, qtRotate  one_leg.cu:339, one_leg_global.cu:97-99       strict
//
// Which candidates are evaluated.  distance_circles evaluates the yaw and the yaw -+ pi and keeps the valid one, or the
// shorter of two equally (in)valid ones.  With sat_D / sat_F the saturated yaws of the two:
//   * behind the coxa (|yaw| beyond limit + pi/2): the direct candidate is "mega-saturated" onto the flipped one with
//     sat_D = yaw -+ pi = sat_F BIT FOR BIT (the same expression): both give the same vector, the result is the flipped one;
//   * in front (the flipped candidate is mega-saturated onto the direct one): sat_F = (yaw -+ pi) +- pi differs from the
//     yaw by rounding.  A valid direct candidate wins by its flag (the mega candidate is always "saturated"); an INVALID
//     one is compared with its twin by strict norms -- the reference's pick is rounding noise, so the twin is evaluated too
//     (same cell, same decisions, its own sincosf / clamp / norm), unless sat_F == yaw bit for bit;
//   * otherwise the two are different configurations: the table's lower bounds order them; a point whose second candidate
//     can still win (0.3 % of a random cloud) is left to the filtered code (doubt).
// Everything else the filtered code decides with strict arithmetic (`limit = angle > coxa_mid`, `sat != angle`) is decided
// here with the same strict comparisons on the same exact angles.
//
// Host and device compile this file; tests/test_xtab_cpu.py holds every point without doubt bit-identical to
// lrm_dist_global / lrm_reach_global on the host, tests/test_gpu_parity.py on the device.
#pragma once
#include "lrm_point_tol.h"

// what the strict value chain reads: 200 bytes, by value in the kernarg segment
struct LrmXtabLeg {
    float inv_rot[9], fwd_rot[9];       // LrmCompiledLeg's
    float cos_body, sin_body, body;
    float cos_pitch, sin_pitch, cos_pitch_rev, sin_pitch_rev;
    float coxa_length, max_coxa, min_coxa, mega_hi, mega_lo, coxa_mid;
    float lim_sc[4];                    // lrm_sincosf(-max_coxa) -> {sin, cos}, lrm_sincosf(-min_coxa) -> {sin, cos}
    float yaw_cs[4];                    // LrmTolLeg::yaw_cs (decision arithmetic)
    float band_base, band_slope, r_outer;
    float pad_[2];
};

inline void lrm_make_xtab_leg(const LrmCompiledLeg& L, const LrmTolLeg& TL, LrmXtabLeg* X) {
    for (int i = 0; i < 9; i++) { X->inv_rot[i] = L.inv_rot[i]; X->fwd_rot[i] = L.fwd_rot[i]; }
    X->cos_body = L.cos_body; X->sin_body = L.sin_body; X->body = L.body;
    X->cos_pitch = L.cos_pitch; X->sin_pitch = L.sin_pitch; X->cos_pitch_rev = L.cos_pitch_rev; X->sin_pitch_rev = L.sin_pitch_rev;
    X->coxa_length = L.coxa_length; X->max_coxa = L.max_coxa; X->min_coxa = L.min_coxa;
    X->mega_hi = L.mega_hi; X->mega_lo = L.mega_lo; X->coxa_mid = L.coxa_mid;
    lrm_sincosf(-L.max_coxa, &X->lim_sc[0], &X->lim_sc[1]); // what finish_finding_closest computes for a candidate clamped to the limit
    lrm_sincosf(-L.min_coxa, &X->lim_sc[2], &X->lim_sc[3]);
    for (int i = 0; i < 4; i++) X->yaw_cs[i] = TL.yaw_cs[i];
    X->band_base = TL.band_base; X->band_slope = TL.band_slope; X->r_outer = TL.r_outer;
    X->pad_[0] = X->pad_[1] = 0.f;
}

// The decisions of lrm_tol_plane_tab (lrm_point_tol.h) at plane point (x, z), x = abscissa - coxa_length: the point's validity
// and WHICH of the cell's two clamp targets wins -- same arithmetic, same bands -- without forming the tolerance vector.
LRM_HD void lrm_xtab_plane(const LrmTolTabView& G, uint32_t code, float x, float z, float band, float tau,
                           float& tx, float& ty, float& tr, bool& valid, uint32_t& doubt) {
    const LrmTabVRow vr = G.vrows[(code >> 10) & 31u];
    const LrmTabRow ra = G.rows[code & 31u], rb = G.rows[(code >> 5) & 31u];
    float vacc;
    {
        const float vx = x - vr.x, vy = z - vr.y;
        vacc = __builtin_fmaf(__builtin_fmaf(vy, vy, vx * vx), vr.gs, vr.c);
    }
    valid = vacc < 0.f;
    const float cpen = valid ? 1.0f : 0.0f; // a corner point only competes when the point is invalid (one_leg.cu:109-116)
    const float ax = x - ra.x, ay = z - ra.y, bx = x - rb.x, by = z - rb.y;
    const float ma = __builtin_fmaf(ay, ay, ax * ax), mb = __builtin_fmaf(by, by, bx * bx);
    const float rsa = LRM_FAST_RSQ(ma), rsb = LRM_FAST_RSQ(mb);
    const float maga = ma * rsa, magb = mb * rsb;
    const float da = ra.r - maga, db = rb.r - magb;
    const float wa = __builtin_fmaf(-ra.chw, maga, __builtin_fmaf(ax, ra.mx, ay * ra.my));
    const float wb = __builtin_fmaf(-rb.chw, magb, __builtin_fmaf(bx, rb.mx, by * rb.my));
    const float cacc = fminf(__builtin_fmaf(-ra.bw, maga, fabsf(wa)), __builtin_fmaf(-rb.bw, magb, fabsf(wb)));
    const float ka = fmaxf(fmaxf(da * da, wa * -1.0e30f), ra.corner * cpen);
    const float kb = fmaxf(fmaxf(db * db, wb * -1.0e30f), rb.corner * cpen);
    const bool wina = ka <= kb;
    const float lo2 = wina ? ka : kb, hi2 = wina ? kb : ka;
    const float a = fabsf(wina ? da : db);
    const float tie_thr = __builtin_fmaf(lo2, 4.0e-6f, __builtin_fmaf(tau, __builtin_fmaf(2.0f, a, tau), lo2));
    tx = wina ? ra.x : rb.x;
    ty = wina ? ra.y : rb.y;
    tr = wina ? ra.r : rb.r;
    const float s = __builtin_fmaf(-tr, wina ? rsa : rsb, 1.0f); // 1 - r / |p - c|
    uint32_t lu = 0;
    lu |= !(fabsf(vacc) > band) ? LRM_TD_REGION : 0u; // (an unanswered cell names validity row 31 = nan)
    lu |= !(cacc > tau) ? LRM_TD_CLAMP : 0u;
    lu |= !(hi2 > tie_thr) ? LRM_TD_TIE : 0u;
    // no target at all (the reference returns the raw point, one_leg.cu:141-142), or a clamp next to the centre of its circle
    // (the reference switches to a fixed direction below |p - c| = 1e-3, one_leg.cu:54-58): left to the filtered code
    lu |= (!(lo2 < 1.0e30f) || !(s > -7.0f)) ? LRM_TD_NONE : 0u;
#if !defined(__HIP_DEVICE_COMPILE__)
    lu |= (code == (uint32_t)LRM_TT_UNANSWERED) ? LRM_TD_AMBIG : 0u;
#endif
    doubt |= lu;
}

// finish_finding_closest<bool> (one_leg.cu:215-278) of ONE yaw candidate, values in reference order, decisions given:
//   a       the point in the coxa frame (strict)
//   (s, c)  sincosf(-sat) of the candidate's saturated yaw;  th = -(limit - sat)
//   cell    the table code of its plane point;  mega: the candidate is mega-saturated (no yaw-limit alternative)
//   wl      the point's offset from the yaw-limit plane `limit` (decision arithmetic): |wl| is d_limit up to rounding
// Returns the candidate's vector in the coxa frame; valid = eval_plane_circles' result (the point passes its circles);
// (tx, ty, tr) = the clamp target that won (for the twin of an invalid candidate, lrm_xtab_twin).
LRM_HD LrmVec3 lrm_xtab_chain(const LrmXtabLeg& X, const LrmTolTabView& G, LrmVec3 p, float s, float c, float th, uint32_t cell, bool mega,
                              float wl, float band, float tau, bool& valid, float& tx, float& ty, float& tr, uint32_t& doubt) {
    float buffer = p.x * s; // cancel_coxa_rotation
    p.x = p.x * c - p.y * s;
    p.y = buffer + p.y * c;
    const LrmVec3 save = p;
    const float x = p.x - X.coxa_length; // eval_plane_circles, one_leg.cu:172
    lrm_xtab_plane(G, cell, x, p.z, band, tau, tx, ty, tr, valid, doubt);
    {   // force_clamp_on_circle onto the winner (one_leg.cu:42-63), then (x, z) -= clamp point (:143-144)
        float cx = x, cy = p.z, d;
        bool v;
        lrm_clamp_on(tx, ty, tr, true, cx, cy, d, v);
        p.x = x - cx;
        p.z = p.z - cy;
    }
    if (valid && !mega) { // one_leg.cu:258-274: the nearer yaw-limit plane wins when it is closer than the in-plane boundary
        const float n2 = __builtin_fmaf(p.x, p.x, __builtin_fmaf(p.y, p.y, p.z * p.z));
        const float dl = fabsf(wl), dl2 = dl * dl;
        const bool near = !(fabsf(n2 - dl2) > band * __builtin_fmaf(2.0f, dl, band));
        if (near || n2 > dl2) { // strict arithmetic decides when `near`, and produces the output
            float s2 = th, c2 = 1.0f; // sincosf(+-0) = (+-0, 1): a candidate clamped to the limit itself
            if (th != 0.f) lrm_sincosf(th, &s2, &c2);
            const float sy = save.x * s2 + save.y * c2;
            LrmVec3 lim = {0.f, sy, 0.f};
            if (lrm_norm3(p) > lrm_norm3(lim)) {
                const float b2 = lim.y * s2;
                lim.y = -lim.x * s2 + lim.y * c2;
                lim.x = lim.x * c2 + b2;
                p = lim;
            }
        }
    }
    buffer = p.y * s; // restore_coxa_rotation
    p.y = -p.x * s + p.y * c;
    p.x = p.x * c + buffer;
    return p;
}

// The twin of an INVALID direct candidate in front of the coxa: the flipped candidate is mega-saturated onto the same meridian
// plane with sat = (yaw -+ pi) +- pi, the yaw up to rounding.  Same cell, same decisions (the plane point moves by less than an
// ulp of its coordinates, far inside every band the first candidate had to clear), no yaw-limit alternative (mega): its own
// sincosf, rotation, strict clamp onto the SAME target, rotation back.
LRM_HD LrmVec3 lrm_xtab_twin(const LrmXtabLeg& X, LrmVec3 p, float sat, float tx, float ty, float tr) {
    float s, c;
    lrm_sincosf(-sat, &s, &c);
    float buffer = p.x * s;
    p.x = p.x * c - p.y * s;
    p.y = buffer + p.y * c;
    const float x = p.x - X.coxa_length;
    float cx = x, cy = p.z, d;
    bool v;
    lrm_clamp_on(tx, ty, tr, true, cx, cy, d, v);
    p.x = x - cx;
    p.z = p.z - cy;
    buffer = p.y * s;
    p.y = -p.x * s + p.y * c;
    p.x = p.x * c + buffer;
    return p;
}

inline thread_local unsigned long long lrm_xtab_host_seconds = 0; // host statistic: points whose twin chain ran

// distance_global + reachability_global (one_leg_global.cu:74-130) of one body-frame point.
// p: in = the point, out = the distance vector.  Returns the flag (distance's validity = the reach mask wherever no decision
// is in doubt, see lrm_reach_from_dist).  doubt != 0: do not use the outputs.
LRM_HD bool lrm_xtab_point(const LrmXtabLeg& X, const LrmTolTabView& G, LrmVec3& p, uint32_t& doubt) {
    // ---- strict: into the coxa frame, the yaw ----
    const float band = __builtin_fmaf(fabsf(p.x) + fabsf(p.y) + fabsf(p.z), X.band_slope, X.band_base);
    const float tau = band * LRM_TOL_TIE;
    LrmVec3 a = lrm_qrot(X.inv_rot, p);
    float buffer = a.x * X.sin_body;
    a.x = a.x * X.cos_body - a.y * X.sin_body;
    a.y = buffer + a.y * X.cos_body;
    a.x -= X.body;
    buffer = a.x * X.sin_pitch;
    a.x = a.x * X.cos_pitch - a.z * X.sin_pitch;
    a.z = buffer + a.z * X.cos_pitch;
    const float ang = lrm_atan2f(a.y, a.x);
    const float ang_flip = (ang > 0) ? ang - LRM_PI_F : ang + LRM_PI_F;
    // ---- decisions, as lrm_tab_point, on the strict coxa-frame point ----
    const float x = a.x, y = a.y, z = a.z;
    const float r = LRM_FAST_SQRT(__builtin_fmaf(y, y, x * x));
    const float cM = X.yaw_cs[0], sM = X.yaw_cs[1], cm = X.yaw_cs[2], sm = X.yaw_cs[3];
    const float uM = __builtin_fmaf(x, cM, y * sM), wM = __builtin_fmaf(y, cM, -(x * sM));
    const float um = __builtin_fmaf(x, cm, y * sm), wm = __builtin_fmaf(y, cm, -(x * sm));
    const uint32_t pat = (lrm_f2u(wM) >> 31) | ((lrm_f2u(uM) >> 30) & 2u) | ((lrm_f2u(wm) >> 29) & 4u) |
                         ((lrm_f2u(um) >> 28) & 8u);
    constexpr uint32_t kLutD = lrm_tol_lut(false), kLutF = lrm_tol_lut(true);
    const uint32_t codeD = (kLutD >> (pat << 1)) & 3u, codeF = (kLutF >> (pat << 1)) & 3u;
    const bool inD = (pat & 5u) == 1u, inF = (pat & 5u) == 4u;
    const float ymin = lrm_min3_aa(wm, um, lrm_min3_aa(wM, uM, 3.0e38f));
    uint32_t lu = !(ymin > band) ? LRM_TD_YAW : 0u; // (no guard around the coxa axis: the values below are the reference's own)
    const bool two = codeD != codeF;
    const bool limD = codeD >= 2u, limF = codeF >= 2u;
    const float wlD = (codeD == 3u) ? wm : wM, wlF = (codeF == 3u) ? wm : wM;
    const float wD = limD ? wlD : 0.f, wF = limF ? wlF : 0.f;
    const float ulD = (codeD == 3u) ? um : uM, ulF = (codeF == 3u) ? um : uM;
    const float urD = lrm_u2f(lrm_f2u(r) ^ (codeD << 31)), urF = lrm_u2f(lrm_f2u(r) ^ (codeF << 31));
    const float uD = limD ? ulD : urD, uF = limF ? ulF : urF;
    const bool far = !(fmaxf(r + X.coxa_length, fabsf(z)) < G.far_limit);
    const bool anyfar = LRM_TOL_ANY(far);
    const float xD = uD - X.coxa_length, xF = uF - X.coxa_length;
    uint32_t cD, cF, sD, sF, fbase;
    float lbD, lbF;
    bool band_doubt;
    if (anyfar) lrm_toltab_lookup2<true>(G, far, band, xD, xF, z, cD, cF, sD, sF, fbase, lbD, lbF, band_doubt);
    else lrm_toltab_lookup2<false>(G, false, band, xD, xF, z, cD, cF, sD, sF, fbase, lbD, lbF, band_doubt);
    lu |= band_doubt ? LRM_TD_YAW : 0u;
    const float bD = __builtin_fmaf(lbD, lbD, wD * wD), bF = __builtin_fmaf(lbF, lbF, wF * wF);
    const bool firstD = inF ? (bD < bF) : (bD <= bF);
    const float b1 = firstD ? bF : bD;
    const bool in0 = firstD ? inD : inF;
    // ---- the first candidate: the strict yaw-limit bookkeeping of finish_finding_closest for candidate (flip, code) ----
    //   direct:  code 0 inside: sat = yaw;        1 mega: sat = yaw -+ pi (= the flipped yaw, the same expression);  2 / 3 clamped
    //   flipped: code 1 inside: sat = yaw -+ pi;  0 mega: sat = (yaw -+ pi) +- pi;                                   2 / 3 clamped
    const float ang_ff = (ang_flip > 0) ? ang_flip - LRM_PI_F : ang_flip + LRM_PI_F;
    const bool flip = !firstD;
    const uint32_t code = flip ? codeF : codeD;
    const bool lim = code >= 2u, mn = code == 3u;
    const bool mega = flip ? (code == 0u) : (code == 1u);
    const float angle = flip ? ang_flip : ang;
    const float limit = (angle > X.coxa_mid) ? X.max_coxa : X.min_coxa;
    const float sat = lim ? (mn ? X.min_coxa : X.max_coxa) : ((code == 1u) ? ang_flip : (flip ? ang_ff : ang));
    float s = mn ? X.lim_sc[2] : X.lim_sc[0], c = mn ? X.lim_sc[3] : X.lim_sc[1];
    if (!lim) lrm_sincosf(-sat, &s, &c);
    const float th = -(limit - sat);
    const uint32_t cell = lrm_toltab_resolve(G, flip ? cF : cD, flip ? sF : sD, fbase);
    const float wl = (limit == X.max_coxa) ? wM : wm;
    bool valid;
    float tx, ty, tr;
    LrmVec3 rv = lrm_xtab_chain(X, G, a, s, c, th, cell, mega, wl, band, tau, valid, tx, ty, tr, lu);
    const bool flag = valid && in0;
    // A DIFFERENT second candidate can still win when the first one is not below its lower bound by more than the tie band
    // (0.3 % of a random cloud): left to the filtered code.
    {
        const float n0 = __builtin_fmaf(rv.x, rv.x, __builtin_fmaf(rv.y, rv.y, rv.z * rv.z));
        lu |= (two && !flag && !(n0 < b1 - tau * __builtin_fmaf(2.0f, LRM_FAST_SQRT(n0), tau))) ? LRM_TD_PICK : 0u;
    }
    // The twin of an invalid direct candidate in front of the coxa: distance_circles' pick (one_leg.cu:334) with
    // res == resflip == false: the strictly shorter DIRECT candidate, else the flipped one.
    const bool twin = !two && firstD && inD && !valid && (ang_ff != ang);
    if (LRM_TOL_ANY(twin)) {
        const LrmVec3 fb = lrm_xtab_twin(X, a, ang_ff, tx, ty, tr);
        const bool take = twin && !(lrm_norm3(rv) < lrm_norm3(fb));
        rv.x = take ? fb.x : rv.x;
        rv.y = take ? fb.y : rv.y;
        rv.z = take ? fb.z : rv.z;
#if !defined(__HIP_DEVICE_COMPILE__)
        lrm_xtab_host_seconds++;
#endif
    }
    // place_over_coxa<Reverse>, z_unrotateInPlace, qtRotate
    buffer = rv.x * X.sin_pitch_rev;
    rv.x = rv.x * X.cos_pitch_rev - rv.z * X.sin_pitch_rev;
    rv.z = buffer + rv.z * X.cos_pitch_rev;
    buffer = rv.x * -X.sin_body;
    rv.x = rv.x * X.cos_body - rv.y * -X.sin_body;
    rv.y = buffer + rv.y * X.cos_body;
    p = lrm_qrot(X.fwd_rot, rv);
    // reachability_global mirrors a point with x < 0 and takes its own atan2f there: within 2 ulp of pi of the flipped yaw
    // (lrm_reach_from_dist); the flag is the mask unless that yaw sits on a limit
    if (lrm_f2u(a.x) >> 31) {
        const float lim_margin = fminf(fabsf(ang_flip - X.max_coxa), fabsf(ang_flip - X.min_coxa));
        lu |= !(lim_margin > 2.0e-6f) ? LRM_TD_YAW : 0u;
    }
    doubt |= lu;
    return flag;
}

// The strict value chain of ONE point whose decisions are known (LRM_MODE_TOL_REL: the tolerance kernel took them, all outside
// their bands, and wrote them into the queue record: see lrm_tab_point<true>): no table, no bands, no doubt -- the reference's
// operations on the winner alone.  p: in = the point, out = the distance vector of distance_global.
LRM_HD void lrm_xtab_replay(const LrmXtabLeg& X, const LrmTabRow* rows, LrmVec3& p, uint32_t info) {
    LrmVec3 a = lrm_qrot(X.inv_rot, p);
    float buffer = a.x * X.sin_body;
    a.x = a.x * X.cos_body - a.y * X.sin_body;
    a.y = buffer + a.y * X.cos_body;
    a.x -= X.body;
    buffer = a.x * X.sin_pitch;
    a.x = a.x * X.cos_pitch - a.z * X.sin_pitch;
    a.z = buffer + a.z * X.cos_pitch;
    const float ang = lrm_atan2f(a.y, a.x);
    const float ang_flip = (ang > 0) ? ang - LRM_PI_F : ang + LRM_PI_F;
    const float ang_ff = (ang_flip > 0) ? ang_flip - LRM_PI_F : ang_flip + LRM_PI_F;
    const LrmTabRow t = rows[(info & 0x8000u) ? ((info >> 5) & 31u) : (info & 31u)];
    const uint32_t code = (info >> 16) & 3u;
    const bool flip = (info & 0x40000u) != 0u, off = (info & 0x80000u) != 0u, flag = (info & 0x100000u) != 0u;
    // an invalid DIRECT candidate inside the yaw range whose flipped candidate is mega-saturated onto it has a twin (lrm_xtab_point);
    // the strict comparisons of finish_finding_closest (one_leg.cu:219-220) on the flipped yaw
    const bool twinp = !flip && code == 0u && ((ang_flip > X.mega_hi) || (ang_flip < X.mega_lo));
    const bool twin = twinp && !flag && (ang_ff != ang);
    const bool lim = code >= 2u, mn = code == 3u;
    const float angle = flip ? ang_flip : ang;
    const float limit = (angle > X.coxa_mid) ? X.max_coxa : X.min_coxa;
    const float sat = lim ? (mn ? X.min_coxa : X.max_coxa) : ((code == 1u) ? ang_flip : (flip ? ang_ff : ang));
    float s = mn ? X.lim_sc[2] : X.lim_sc[0], c = mn ? X.lim_sc[3] : X.lim_sc[1];
    // This function is a latency chain run by one wave at the end of its workgroup (the workgroup's LDS and wave slots wait for it):
    // the twin's value chain -- independent of the candidate's once the yaw is known -- is written next to it, not behind a branch,
    // so that the two sincosf / clamp chains interleave (a wave of short vectors nearly always holds a lane that needs the twin).
    float s1, c1;
    if (!lim) lrm_sincosf(-sat, &s, &c);
    lrm_sincosf(-ang_ff, &s1, &c1);
    LrmVec3 q = a, q1 = a;
    buffer = q.x * s; // cancel_coxa_rotation
    q.x = q.x * c - q.y * s;
    q.y = buffer + q.y * c;
    float buffer1 = q1.x * s1;
    q1.x = q1.x * c1 - q1.y * s1;
    q1.y = buffer1 + q1.y * c1;
    {   // the twin: force_clamp_on_circle onto the same target (lrm_xtab_twin, written out)
        const float x1 = q1.x - X.coxa_length;
        float cx = x1, cy = q1.z, d;
        bool v;
        lrm_clamp_on(t.x, t.y, t.r, true, cx, cy, d, v);
        q1.x = x1 - cx;
        q1.z = q1.z - cy;
    }
    if (LRM_TOL_ANY(off)) { // the offset from the yaw-limit plane (one_leg.cu:258-274 decided `d_clamped > d_limit`)
        const float th = -(limit - sat);
        float s2 = th, c2 = 1.0f;
        if (th != 0.f) lrm_sincosf(th, &s2, &c2);
        const float sy = q.x * s2 + q.y * c2;
        LrmVec3 l = {0.f, sy, 0.f};
        const float b2 = l.y * s2;
        l.y = -l.x * s2 + l.y * c2;
        l.x = l.x * c2 + b2;
        // force_clamp_on_circle onto the winner (one_leg.cu:42-63), (x, z) -= clamp point (:143-144)
        const float x = q.x - X.coxa_length;
        float cx = x, cy = q.z, d;
        bool v;
        lrm_clamp_on(t.x, t.y, t.r, true, cx, cy, d, v);
        q.x = off ? l.x : x - cx;
        q.y = off ? l.y : q.y;
        q.z = off ? l.z : q.z - cy;
    } else {
        const float x = q.x - X.coxa_length;
        float cx = x, cy = q.z, d;
        bool v;
        lrm_clamp_on(t.x, t.y, t.r, true, cx, cy, d, v);
        q.x = x - cx;
        q.z = q.z - cy;
    }
    buffer = q.y * s; // restore_coxa_rotation
    q.y = -q.x * s + q.y * c;
    q.x = q.x * c + buffer;
    buffer1 = q1.y * s1;
    q1.y = -q1.x * s1 + q1.y * c1;
    q1.x = q1.x * c1 + buffer1;
    {   // distance_circles' pick (one_leg.cu:334) with res == resflip == false: the strictly shorter DIRECT candidate, else the flipped one
        const bool take = twin && !(lrm_norm3(q) < lrm_norm3(q1));
        q.x = take ? q1.x : q.x;
        q.y = take ? q1.y : q.y;
        q.z = take ? q1.z : q.z;
    }
    buffer = q.x * X.sin_pitch_rev;
    q.x = q.x * X.cos_pitch_rev - q.z * X.sin_pitch_rev;
    q.z = buffer + q.z * X.cos_pitch_rev;
    buffer = q.x * -X.sin_body;
    q.x = q.x * X.cos_body - q.y * -X.sin_body;
    q.y = buffer + q.y * X.cos_body;
    p = lrm_qrot(X.fwd_rot, q);
}
