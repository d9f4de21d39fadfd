// gnsscorr_ctx.h -- host-side context object behind the opaque gnsscorr_ctx.
#pragma once

#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <cstring>
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

    // ingest: copy stream + two pinned staging buffers (and device staging for packed formats); ev_in marks
    // the last transfer, the compute stream waits on it before it reads the ring
    hipStream_t stream_in = nullptr;
    int8_t *pin[2] = {nullptr, nullptr};
    uint8_t *dstage[2] = {nullptr, nullptr};
    hipEvent_t ev_pin[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    int pin_next = 0;
    hipEvent_t ev_in = nullptr;
    bool in_pending = false;

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

    // tracking.  The planner (a short sequential NCO chain per channel) runs one batch ahead on
    // its own stream: plan entries and the chained state are double buffered, so batch k+1 is
    // planned while batch k is correlated.  A look-ahead plan is dropped when the caller changes
    // the state or the batch length.
    GcTrkState *dstate2[2] = {nullptr, nullptr};   // ping-pong; cur = index of the committed state
    int state_cur = 0;
    GcTrkPlan *dplan2[2] = {nullptr, nullptr};
    // claims of the batch being planned (discovery pass -> chain), two buffers: while the chain of one batch reads
    // its claims the discovery pass of the NEXT batch fills the other one from the same input state, on its own stream
    int *dspec2[2] = {nullptr, nullptr};
    unsigned short *detab = nullptr;               // [unit][GC_EDGTAB] start samples of the batch's chip edges (main stream: trk_edges -> trk_corr)
    hipStream_t stream4 = nullptr;                 // discovery-ahead stream
    hipEvent_t ev_spec = nullptr, ev_chain = nullptr;
    bool spec_pending = false;                     // stream4 has work whose end ev_spec marks
    bool spec_ahead_valid = false;                 // dspec2[spec_ahead_buf] holds claims for the batch that starts at spec_ahead_state
    int spec_ahead_buf = 0, spec_ahead_nepoch = 0;
    int spec_last_buf = 0, spec_last_units = 0;     // (tools/debug) the claims the last planned batch used
    const void *spec_ahead_state = nullptr;
    int plan_slot = 0;                             // slot the next trk_run consumes
    bool ahead_valid = false;                      // dplan2[plan_slot] already planned (look-ahead)
    int ahead_nepoch = 0;
    bool state_touched = true;                     // set_state since the last run: do not look ahead
    hipStream_t stream2 = nullptr;                 // planner stream
    hipStream_t stream3 = nullptr;                 // finish stream
    hipEvent_t ev_plan[2] = {nullptr, nullptr};    // plan of slot s finished
    hipEvent_t ev_used[2] = {nullptr, nullptr};    // expand consumed slot s
    // per-unit constants, one set per plan slot: the planner stream expands batch k+1 while batch k is correlated
    GcTrkUnit *dunit2[2] = {nullptr, nullptr};
    GcRound *drounds2[2] = {nullptr, nullptr};     // [unit][nseg][GC_MAXR]
    GcUnitSegs *dsegs2[2] = {nullptr, nullptr};    // [unit]: the unit's carrier / code NCO piece tables
    int *dnco_overflow = nullptr;
    int *dring_viol = nullptr;                     // planned periods outside what the ring holds, since the last fetch
    // closed loop (gnsscorr_trk_run_loop): per channel loop state, one log row per period; the step buffers hold one
    // filter interval per channel (GC_STEP_KMAX periods at most): unit constants, NCO tables, rounds, partial sums
    gnsscorr_loop_t *dloop = nullptr;              // [nch]
    GcStepMeta *dstep_meta = nullptr;              // [nch]
    GcTrkUnit *dstep_unit = nullptr;               // [nch][GC_STEP_KMAX]
    GcUnitSegs *dstep_segs = nullptr;
    GcRound *dstep_rounds = nullptr;               // [nch][GC_STEP_KMAX][step_nseg][4]
    int *dstep_partial = nullptr;                  // [nch][GC_STEP_KMAX][step_nseg][2*ntap]
    int step_nseg = 0;
    unsigned *hostflags = nullptr;                 // pinned, device-visible: [0] channels whose run is over, [1] some channel has its nav bit synchronised
    unsigned *hostflags_dev = nullptr;
    bool loop_sync_hint = false;                   // some channel had its nav bit synchronised when last seen
    int loop_kmax = 1;                             // largest loopms among the channels' loop states (gnsscorr_loop_set)
    gnsscorr_trklog_t *dlooplog = nullptr;         // [nch][looplog_cap]
    int *dloopdone = nullptr;                      // [nch]
    size_t looplog_cap = 0;
    int last_loop_nper = 0;                        // > 0: the last run was a closed-loop one of that many periods                  // units whose NCO tables overflowed since the last fetch
    int *dnsamp2[2] = {nullptr, nullptr};
    int last_slot = 0;                             // slot of the last completed trk_run
    size_t plan_cap = 0;
    double *dcorrI = nullptr, *dcorrQ = nullptr, *dsumI = nullptr, *dsumQ = nullptr;
    unsigned long long *dfinish = nullptr;     // batch-sum scratch of trk_finish
    int *dpartial2[2] = {nullptr, nullptr};    // per slot: [ch][epoch][segment][2*ntap] int32 partial sums
    hipEvent_t ev_corr[2] = {nullptr, nullptr};   // correlator of slot s finished (main stream)
    hipEvent_t ev_fin[2] = {nullptr, nullptr};    // finish of slot s done (finish stream): outputs valid
    bool fin_pending[2] = {false, false};         // ev_fin[s] has been recorded
    int nseg = 1;
    int last_nepoch = 0;

    // acquisition
    GcAcqWork *acq = nullptr;

    // timing
    int timing = 0;                                // 0: off, 1: every kernel, 2: only the two correlator kernels (trk_corr, acq_corr)
    std::map<std::string, GcTimer> timers;
};

// RAII helper: brackets one kernel launch with HIP events when timing is on.
struct GcTimed {
    gnsscorr_ctx *ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    hipStream_t st;
    GcTimed(gnsscorr_ctx *c, const char *n, hipStream_t s = nullptr) : ctx(c), name(n), st(s ? s : c->stream)
    {
        if (!ctx->timing) return;
        if (ctx->timing == 2 && strcmp(n, "trk_corr") != 0 && strcmp(n, "acq_corr") != 0) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        hipEventRecord(a, st);
    }
    ~GcTimed()
    {
        if (!a) return;
        hipEventRecord(b, st);
        ctx->timers[name].pending.emplace_back(a, b);
    }
};

void gc_acq_free(gnsscorr_ctx *ctx);
int gc_ingest_fence(gnsscorr_ctx *ctx);      // orders the compute stream behind the last ring transfer
