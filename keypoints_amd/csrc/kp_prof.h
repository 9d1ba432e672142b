// In-library kernel timing (bench.py's roofline leg): HIP events recorded on the launch stream right around a kernel
// family's launches; kp_prof_collect() synchronises them after the timed region.  Storage lives in kp_api.hip.
#pragma once
#include <mutex>
#include <vector>
#include "kp_common.h"

struct ProfRec { int family; double flops, bytes; hipEvent_t e0, e1; };
extern bool g_prof_on;
extern std::vector<ProfRec> g_prof;
extern std::mutex g_prof_mu;   // guards g_prof and the event pool: launches may come from several host threads
hipEvent_t kp_prof_event();
void kp_prof_event_release(hipEvent_t e);

struct ProfScope {
    bool on; ProfRec r; hipStream_t st;
    double counted;
    ProfScope(int family, double flops, hipStream_t s, double bytes = 0.0) : on(g_prof_on && !kp_query_mode()), st(s), counted(bytes) {
        kp_count_bytes(bytes);   // (step-level byte accounting, kp_prof_enable(2): no events)
        if (!on) return;
        r.family = family; r.flops = flops; r.bytes = bytes; r.e0 = kp_prof_event(); r.e1 = kp_prof_event();
        (void)hipEventRecord(r.e0, st);
    }
    void cancel() {   // nothing was launched under this scope: give the events back
        kp_count_bytes(-counted, -1); counted = 0.0;
        if (!on) return;
        kp_prof_event_release(r.e0); kp_prof_event_release(r.e1);
        on = false;
    }
    void done(int family) {
        if (!on) return;
        r.family = family;
        (void)hipEventRecord(r.e1, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};
