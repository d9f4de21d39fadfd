// gnsscorr_ctx.h -- host-side context object behind the opaque gnsscorr_ctx.
#pragma once

#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "gnsscorr_internal.h"

struct GcRing {
    int8_t  *mem = nullptr;
    bool     owned = false;
    int      dtype = 0;
    uint64_t ringlen = 0;     // samples
    uint64_t wrpos = 0;       // samples written so far (fendbuffsize*buffcnt)
};

struct GcTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int launches = 0;
};

struct GcAcqWork;   // gnsscorr_acq.hip

struct gnsscorr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mtx;

    GcRing ring[2];

    // channels
    int nch = 0;
    std::vector<GcChan> hchan;
    std::vector<gnsscorr_chan_t> hdesc;         // host copies (code/freq/corrp pointers re-targeted)
    std::vector<std::vector<short>> hcode;
    std::vector<std::vector<double>> hfreq;
    std::vector<std::vector<int>> hcorrp;
    GcChan *dchan = nullptr;
    int8_t *dcodes = nullptr;
    double *dfreqs = nullptr;
    int ntap = 0, smax_max = 0, max_n = 0;

    // tracking
    GcTrkState *dstate = nullptr;
    GcTrkPlan *dplan = nullptr;
    GcTrkUnit *dunit = nullptr;
    size_t plan_cap = 0;
    double *dcorrI = nullptr, *dcorrQ = nullptr, *dsumI = nullptr, *dsumQ = nullptr;
    int *dnsamp = nullptr;
    int *dpartial = nullptr;       // [ch][epoch][segment][2*ntap] int32 partial sums
    int nseg = 1;
    int last_nepoch = 0;

    // acquisition
    GcAcqWork *acq = nullptr;

    // timing
    bool timing = false;
    std::map<std::string, GcTimer> timers;
};

// RAII helper: brackets one kernel launch with HIP events when timing is on.
struct GcTimed {
    gnsscorr_ctx *ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    GcTimed(gnsscorr_ctx *c, const char *n) : ctx(c), name(n)
    {
        if (!ctx->timing) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        hipEventRecord(a, ctx->stream);
    }
    ~GcTimed()
    {
        if (!a) return;
        hipEventRecord(b, ctx->stream);
        ctx->timers[name].pending.emplace_back(a, b);
    }
};

void gc_acq_free(gnsscorr_ctx *ctx);
