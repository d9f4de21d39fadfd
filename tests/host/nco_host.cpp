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

// end values only (what the planner chains)
void nco_chain(double phi0, double freq, double ti, int n, int len, double coff, int smax, double ci,
               double *prem, double *rem)
{
    GcNoEmit ne;
    *prem = gc_carrier_prem(gc_carrier_walk(gc_carrier_phis(phi0), gc_carrier_ps(freq, ti), n, ne));
    *rem = gc_code_rem(gc_code_walk(gc_code_start(coff, smax, ci, len), ci, len, n + 2 * smax, ne), smax, ci);
}

}
