// generated from gnsscorr_nco.h's gc_code_period_t for section timing (ubench only)
template <int ITOP, class Fill, class Emit>
GC_HD bool gc_code_period_prof(long long *T, const GcCodePlan &P, double remcode, int nt, Fill &fill, double *remcode_out, Emit &emit)
{
    GC_FP_STRICT
    long long t0 = __builtin_readcyclecounter(), t1;
#define TK(i) do { t1 = __builtin_readcyclecounter(); T[i] += t1 - t0; t0 = t1; } while (0)
    const GcNcoFast &f = P.f;
    const double ci = f.s, dlen = P.dlen;
    // ---- start value (ref :613-614) and head: c0 + j d_top < len for j <= q
    double cs = remcode - P.smaxci;
    const double fl = cs < 0.0 ? -1.0 : 0.0;
    if (!(cs >= -dlen && cs < dlen)) return false;
    const double c0 = cs - fl * dlen;
    if ((int)((gc_d2u(c0) >> 52) & 0x7FF) != f.ex0 + ITOP || !(c0 < dlen)) return false;
    const double dtop = f.d[ITOP];
    double y;
    int j;
    {
        const double R = P.limtop - c0;
        double q = floor(R * f.inv[ITOP]);
        const double r = fma(-q, dtop, R);
        q += r < 0.0 ? -1.0 : (r >= dtop ? 1.0 : 0.0);
        if (!(q >= 0.0 && q < (double)(nt - 2))) return false;
        y = fma(q + 1.0, dtop, c0) - dlen;          // first sample at or above len, wrapped (exact)
        j = (int)q + 1;
        emit(0, c0, dtop, j, 0);
    }
    TK(0);
    // ---- next to zero: the reference's own additions up to the table
    const double b0 = gc_u2d((uint64_t)f.ex0 << 52);
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const bool lit = y < b0;
        if (lit) emit(j, y, 0.0, 1, 1);
        const double yl = y + ci;
        y = lit ? yl : y;
        j += lit ? 1 : 0;
    }
    if (!(y >= b0) || j >= nt - 2) return false;
    TK(1);
    // ---- climb: crossings (one boundary per lane), then the chain
    GcCertCtx c;
    c.a0 = y;
    c.sabs = ci;
    c.inv = f.inv_s;
    c.n = nt - j;
    c.exact = P.exact;
    c.ex0 = f.ex0;
    const int i0 = (int)((gc_d2u(y) >> 52) & 0x7FF) - f.ex0;
    if (i0 < 0 || i0 > 1) return false;
    int K[GC_NB + 1];
    if (!fill(K, c, i0, ITOP, dlen)) return false;
    TK(2);
    if (K[GC_NB] >= c.n) return false;              // (the period must end in the tail)
    gc_code_climb_lean<ITOP>(f, K, i0, P.it, ci, &y, j, emit);
    j += K[GC_NB];
    TK(3);
    if (j >= nt || !(y >= dlen)) return false;
    // ---- second wrap and tail
    y = y - dlen;
    int t = nt - j;
    for (; t >= 4; t -= 4) {                        // (the reference's own additions, four per trip)
        emit(nt - t, y, 0.0, 1, 2);
        y = y + ci;
        emit(nt - t + 1, y, 0.0, 1, 2);
        y = y + ci;
        emit(nt - t + 2, y, 0.0, 1, 2);
        y = y + ci;
        emit(nt - t + 3, y, 0.0, 1, 2);
        y = y + ci;
    }
    for (; t > 0; t--) {
        emit(nt - t, y, 0.0, 1, 2);
        y = y + ci;
    }
    *remcode_out = y - P.smaxci;
    TK(4);
    return true;
}

