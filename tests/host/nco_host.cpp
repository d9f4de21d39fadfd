// Host-side harness around the product's NCO emulation (csrc/gnsscorr_nco.h), built by
// tests/test_nco_host.py with g++ -ffp-contract=off: the same header the kernels use, here
// evaluated on the CPU so that it can be compared with the oracle's literal loops without a GPU.
#include "../../erlangnetwork-gnsslib-sdr_amd/csrc/gnsscorr_nco.h"

extern "C" {

// LUT index of every sample + phase remainder, through the segment table; returns the number of
// pieces, or -1 when the table overflowed `cap`
int nco_carrier(double phi0, double freq, double ti, int n, int cap, int *idx, double *prem)
{
    int k0[256];
    GcCarSeg seg[256];
    if (cap > 256) cap = 256;
    GcCarTable t{k0, seg, cap, 0, 0};
    const double xn = gc_carrier_walk(gc_carrier_phis(phi0), gc_carrier_ps(freq, ti), n, t);
    *prem = gc_carrier_prem(xn);
    if (t.overflow) return -1;
    for (int k = 0; k < n; k++) idx[k] = gc_carrier_idx_at(k0, seg, t.n, k);
    return t.n;
}

// chip index (wrapped) of every replica position + returned remainder
int nco_code(int len, double coff, int smax, double ci, int n, int cap, int *chip, double *rem)
{
    GcCodeSeg seg[256];
    if (cap > 256) cap = 256;
    GcCodeTable t{seg, cap, 0, 0};
    const int nt = n + 2 * smax;
    const double cend = gc_code_walk(gc_code_start(coff, smax, ci, len), ci, len, nt, t);
    *rem = gc_code_rem(cend, smax, ci);
    if (t.overflow) return -1;
    for (int j = 0; j < nt; j++) chip[j] = gc_code_chip_at(seg, t.n, j, nullptr);
    return t.n;
}

// the same through the tabulated fast walkers (what the device planner and expansion run)
int nco_carrier_fast(double phi0, double freq, double ti, int n, int cap, int *idx, double *prem)
{
    int k0[256];
    GcCarSeg seg[256];
    if (cap > 256) cap = 256;
    GcCarTable t{k0, seg, cap, 0, 0};
    GcNcoFast f;
    gc_fast_init(f, gc_carrier_ps(freq, ti));
    const double xn = gc_fast_carrier_walk(f, gc_carrier_phis(phi0), n, t);
    GcNcoFast fp;
    gc_fast_init(fp, -GC_NCO_DPI);
    *prem = gc_fast_prem(fp, xn);
    if (t.overflow) return -1;
    for (int k = 0; k < n; k++) idx[k] = gc_carrier_idx_at(k0, seg, t.n, k);
    return t.n;
}

int nco_code_fast(int len, double coff, int smax, double ci, int n, int cap, int *chip, double *rem)
{
    GcCodeSeg seg[256];
    if (cap > 256) cap = 256;
    GcCodeTable t{seg, cap, 0, 0};
    GcNcoFast f;
    gc_fast_init(f, ci);
    const int nt = n + 2 * smax;
    const double cend = gc_fast_code_walk(f, gc_code_start(coff, smax, ci, len), len, nt, t);
    *rem = gc_code_rem(cend, smax, ci);
    if (t.overflow) return -1;
    for (int j = 0; j < nt; j++) chip[j] = gc_code_chip_at(seg, t.n, j, nullptr);
    return t.n;
}

// the planner's chain step exactly as trk_plan_kernel writes it (reciprocal divisions, fast start)
void nco_chain_fast(double phi0, double freq, double ti, double f_sf, double codefreq, int len, double coff, int smax,
                    int *n_out, double *prem, double *rem)
{
    const double ci = ti * codefreq, spc = codefreq / f_sf, ps = gc_carrier_ps(freq, ti);
    GcNcoFast fcar, fcode, fprem;
    gc_fast_init(fcar, ps);
    gc_fast_init(fcode, ci);
    gc_fast_init(fprem, -GC_NCO_DPI);
    const double yspc = 1.0 / spc, ydpi = 1.0 / GC_NCO_DPI, smaxci = (double)smax * ci;
    const int n = (int)gc_div_y((double)len - coff, spc, yspc);
    GcNoEmit ne;
    const double phis = gc_div_y(phi0 * GC_NCO_CDIV, GC_NCO_DPI, ydpi);
    *prem = gc_fast_prem(fprem, gc_fast_carrier_walk(fcar, phis, n, ne));
    *rem = gc_fast_code_walk(fcode, gc_code_start_fast(coff, smaxci, len), len, n + 2 * smax, ne) - smaxci;
    *n_out = n;
}

// the certified-crossing planner path; *used = 1 when the certified path produced the values (else fallback)
void nco_chain_cert(double phi0, double freq, double ti, double f_sf, double codefreq, int len, double coff, int smax,
                    int *n_out, double *prem, double *rem, int *used)
{
    const double ci = ti * codefreq, spc = codefreq / f_sf, ps = gc_carrier_ps(freq, ti);
    GcNcoFast fcar, fcode, fprem;
    gc_fast_init(fcar, ps);
    gc_fast_init(fcode, ci);
    gc_fast_init(fprem, -GC_NCO_DPI);
    const double yspc = 1.0 / spc, ydpi = 1.0 / GC_NCO_DPI, smaxci = (double)smax * ci;
    const int n = (int)gc_div_y((double)len - coff, spc, yspc);
    GcNoEmit ne;
    int K[GC_NB + 2];
    const double phis = gc_div_y(phi0 * GC_NCO_CDIV, GC_NCO_DPI, ydpi);
    double xn, cend;
    *used = 0;
    if (gc_plan_carrier_walk(fcar, phis, n, K, &xn)) *used |= 1;
    else xn = gc_fast_carrier_walk(fcar, phis, n, ne);
    *prem = gc_fast_prem(fprem, xn);
    const double c0 = gc_code_start_fast(coff, smaxci, len);
    if (gc_plan_code_walk(fcode, c0, len, n + 2 * smax, K, &cend)) *used |= 2;
    else cend = gc_fast_code_walk(fcode, c0, len, n + 2 * smax, ne);
    *rem = cend - smaxci;
    *n_out = n;
}

// the shape-specialised code period step (device: trk_plan_kernel); returns 1 when it applied
int nco_code_period(double ti, double codefreq, int len, double remcode, int smax, int nt, double *rem)
{
    GcCodePlan P;
    gc_code_plan_init(P, ti * codefreq, len, smax);
    GcFillLoop fill;
    return gc_code_period(P, remcode, nt, fill, rem) ? 1 : 0;
}

int nco_carrier_period(double ti, double freq, double remcarr, int n, double *prem)
{
    GcCarPlan P;
    gc_car_plan_init(P, gc_carrier_ps(freq, ti));
    GcFillLoop fill;
    return gc_carrier_period(P, remcarr, n, fill, prem) ? 1 : 0;
}

// period steps with table emission (what the closed-loop kernel runs): every index from the tables
int nco_period_tables(double ti, double freq, double remcarr, double codefreq, int len, double remcode, int smax, int n,
                      int *idx, int *chip, double *prem, double *rem)
{
    int k0[GC_NB * 3];
    GcCarSeg cseg[GC_NB * 3];
    GcCodeSeg dseg[GC_NB * 3];
    GcCarTable ct{k0, cseg, GC_NB * 3, 0, 0};
    GcCodeTable dt{dseg, GC_NB * 3, 0, 0};
    GcCodePlan PC;
    GcCarPlan PK;
    gc_code_plan_init(PC, ti * codefreq, len, smax);
    gc_car_plan_init(PK, gc_carrier_ps(freq, ti));
    GcFillLoop fill;
    int r = 0;
    if (gc_carrier_period(PK, remcarr, n, fill, prem, ct) && !ct.overflow) {
        r |= 1;
        for (int k = 0; k < n; k++) idx[k] = gc_carrier_idx_at(k0, cseg, ct.n, k);
    }
    if (gc_code_period(PC, remcode, n + 2 * smax, fill, rem, dt) && !dt.overflow) {
        r |= 2;
        for (int j = 0; j < n + 2 * smax; j++) chip[j] = gc_code_chip_at(dseg, dt.n, j, nullptr);
    }
    return r;
}

// The batch planner's steps on claims (gnsscorr_plan.hip: trk_spec_kernel discovers, trk_plan2_kernel evaluates): state after every
// period; hits[0]/[1]: periods the code / carrier evaluation served, the rest go to the certified steps
// ([2]/[3]) and the walkers ([4]/[5]).  shift_*: added to the discovering run's start values.
void nco_claims_chain(double ti, double f_sf, double freq, double codefreq, int len, int smax, double remcode0, double remcarr0,
                      int nepoch, double shift_code, double shift_car, double *rem_out, double *prem_out, int *n_out, int *hits)
{
    GC_FP_STRICT
    const double ci = ti * codefreq, spc = codefreq / f_sf, ps = gc_carrier_ps(freq, ti), dlen = (double)len;
    GcCodePlan PC;
    GcCarPlan PK;
    gc_code_plan_init(PC, ci, len, smax);
    gc_car_plan_init(PK, ps);
    GcNcoFast fcode = PC.f, fcar = PK.f;
    GcCarStepC CK;
    gc_car_stepc_init(CK, PK, (int)(f_sf * 1e-3) + 16);
    const double smaxci = (double)smax * ci, ydpi = 1.0 / GC_NCO_DPI;
    GcNoEmit ne;
    double remcode = remcode0, remcarr = remcarr0;
    for (int i = 0; i < 6; i++) hits[i] = 0;
    for (int e = 0; e < nepoch; e++) {
        GcCodeClaims cc;
        GcCarClaims ck;
        cc.tag = 0;
        ck.tag = 0;
        {
            double rc, rk, dummy;
            int ns;
            gc_spec_start(remcode0, remcarr0, ci, spc, ps, dlen, e, &rc, &rk, &ns);
            if (ns > 0) {
                gc_code_claims<true>(PC, rc + shift_code, ns + 2 * smax, cc, &dummy);
                gc_carrier_claims_step<true>(PK, CK, rk + shift_car, ns, ck, &dummy);
            }
        }
        const int n = (int)((dlen - remcode) / spc);
        n_out[e] = n;
        double r;
        GcFillLoop fill;
        if (gc_carrier_claims_step<false>(PK, CK, remcarr, n, ck, &r)) hits[1]++;
        else if (gc_carrier_period(PK, remcarr, n, fill, &r)) hits[3]++;
        else {
            hits[5]++;
            r = gc_fast_prem(PK.fprem, gc_fast_carrier_walk(fcar, gc_div_y(remcarr * GC_NCO_CDIV, GC_NCO_DPI, ydpi), n, ne));
        }
        remcarr = r;
        if (gc_code_claims<false>(PC, remcode, n + 2 * smax, cc, &r)) hits[0]++;
        else if (gc_code_period(PC, remcode, n + 2 * smax, fill, &r)) hits[2]++;
        else {
            hits[4]++;
            r = gc_fast_code_walk(fcode, gc_code_start_fast(remcode, smaxci, len), len, n + 2 * smax, ne) - smaxci;
        }
        remcode = r;
        rem_out[e] = remcode;
        prem_out[e] = remcarr;
    }
}

// end values only (what the planner chains)
void nco_chain(double phi0, double freq, double ti, int n, int len, double coff, int smax, double ci,
               double *prem, double *rem)
{
    GcNoEmit ne;
    *prem = gc_carrier_prem(gc_carrier_walk(gc_carrier_phis(phi0), gc_carrier_ps(freq, ti), n, ne));
    *rem = gc_code_rem(gc_code_walk(gc_code_start(coff, smax, ci, len), ci, len, n + 2 * smax, ne), smax, ci);
}

}
