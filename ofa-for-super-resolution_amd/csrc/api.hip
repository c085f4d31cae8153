// api.hip -- version / error reporting of the C ABI (include/ofasr.h).
#include <cxxabi.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include "ofasr_common.h"

namespace ofasr {
static thread_local char g_err[512] = {0};
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- launch sites, counters and the optional event profile (see OFASR_LAUNCH in ofasr_common.h) ----------------
int g_profile_on = 0;
namespace {
std::mutex g_prof_mu;
LaunchSite* g_sites = nullptr;
struct ProfRec {
    ProfEvents ev;   // first member: prof_begin hands out &ev
    LaunchSite* site;
    double bytes, flops;
};
std::vector<ProfRec*> g_recs;       // launches bracketed since the last ofasr_profile_enable(1)
std::vector<hipEvent_t> g_ev_pool;  // events handed back by ofasr_profile_read
thread_local double t_note_bytes = 0.0, t_note_flops = 0.0;
std::string g_report;

hipEvent_t take_event() {
    if (!g_ev_pool.empty()) {
        hipEvent_t e = g_ev_pool.back();
        g_ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return e;
}
}  // namespace

LaunchSite* register_site(const void* host_fn, const char* fallback) {
    // the device symbol of the host stub, demangled, without "void ofasr::" and the parameter list:
    //   "dw_mfma_kernel<ofasr::bf16_t, 7, false, true, true>"
    std::string n;
    const char* mangled = hipKernelNameRefByPtr(host_fn, nullptr);
    if (mangled && mangled[0]) {
        int status = 0;
        char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
        n = (status == 0 && dem) ? dem : mangled;
        free(dem);
        int depth = 0;   // cut the parameter list: the last top-level '(' of the demangled name
        size_t cut = std::string::npos;
        for (size_t i = 0; i < n.size(); ++i) {
            if (n[i] == '<') ++depth;
            else if (n[i] == '>') --depth;
            else if (n[i] == '(' && depth == 0) { cut = i; break; }
        }
        if (cut != std::string::npos) n.erase(cut);
        if (n.compare(0, 5, "void ") == 0) n.erase(0, 5);
        if (n.compare(0, 7, "ofasr::") == 0) n.erase(0, 7);
    } else {
        (void)hipGetLastError();
        n = fallback ? fallback : "?";
    }
    LaunchSite* s = new LaunchSite{strdup(n.c_str()), 0ull, nullptr};
    std::lock_guard<std::mutex> lk(g_prof_mu);
    s->next = g_sites;
    g_sites = s;
    return s;
}

void prof_note(double bytes, double flops) {
    t_note_bytes = bytes;
    t_note_flops = flops;
}

ProfEvents* prof_begin(LaunchSite* s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t a = take_event(), b = take_event();
    if (!a || !b) {
        if (a) g_ev_pool.push_back(a);
        return nullptr;
    }
    ProfRec* r = new ProfRec{{a, b}, s, t_note_bytes, t_note_flops};
    t_note_bytes = t_note_flops = 0.0;
    g_recs.push_back(r);
    return &r->ev;
}
}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT long long ofasr_debug_launch_count(const char* substr) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    long long n = 0;
    for (LaunchSite* s = g_sites; s; s = s->next)
        if (!substr || !substr[0] || strstr(s->name, substr)) n += (long long)__atomic_load_n(&s->count, __ATOMIC_RELAXED);
    return n;
}

OFASR_EXPORT void ofasr_debug_reset_launch_counts(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (LaunchSite* s = g_sites; s; s = s->next) __atomic_store_n(&s->count, 0ull, __ATOMIC_RELAXED);
}

// one line per launch site that was ever reached: "count<TAB>name"
OFASR_EXPORT const char* ofasr_debug_launch_table(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_report.clear();
    for (LaunchSite* s = g_sites; s; s = s->next) {
        char head[32];
        snprintf(head, sizeof(head), "%llu\t", __atomic_load_n(&s->count, __ATOMIC_RELAXED));
        g_report += head;
        g_report += s->name;
        g_report += "\n";
    }
    return g_report.c_str();
}

OFASR_EXPORT int ofasr_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int was = g_profile_on;
    if (on) {
        // events for ~5 steps of the training path up front: creating them launch by launch slows the host enough to
        // change how the two streams overlap, i.e. the regime that is being measured
        while (g_ev_pool.size() < 6144) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            g_ev_pool.push_back(e);
        }
    }
    g_profile_on = on ? 1 : 0;
    return was;
}

// Blocks until the bracketed launches have finished (this is the one synchronising entry point of the library: a
// measurement tool, never on the product path), then returns one line per kernel
//   "name<TAB>launches<TAB>total_us<TAB>bytes<TAB>flops"
// and forgets the records.  The string stays valid until the next call.
OFASR_EXPORT const char* ofasr_profile_read(void) {
    std::vector<ProfRec*> recs;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        recs.swap(g_recs);
    }
    struct Agg { double n = 0, us = 0, bytes = 0, flops = 0; };
    std::vector<std::pair<LaunchSite*, Agg>> agg;
    for (ProfRec* r : recs) {
        float ms = 0.f;
        const bool ok = hipEventSynchronize(r->ev.t1) == hipSuccess && hipEventElapsedTime(&ms, r->ev.t0, r->ev.t1) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        Agg* a = nullptr;
        for (auto& kv : agg)
            if (kv.first == r->site) a = &kv.second;
        if (!a) {
            agg.push_back({r->site, Agg{}});
            a = &agg.back().second;
        }
        if (ok) {
            a->n += 1;
            a->us += 1e3 * (double)ms;
            a->bytes += r->bytes;
            a->flops += r->flops;
        }
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRec* r : recs) {
        g_ev_pool.push_back(r->ev.t0);
        g_ev_pool.push_back(r->ev.t1);
        delete r;
    }
    g_report.clear();
    for (auto& kv : agg) {
        char line[160];
        snprintf(line, sizeof(line), "\t%.0f\t%.3f\t%.0f\t%.0f\n", kv.second.n, kv.second.us, kv.second.bytes,
                 kv.second.flops);
        g_report += kv.first->name;
        g_report += line;
    }
    return g_report.c_str();
}

OFASR_EXPORT int ofasr_version(void) { return OFASR_VERSION; }
OFASR_EXPORT const char* ofasr_last_error_string(void) { return ofasr::g_err; }
OFASR_EXPORT const char* ofasr_status_string(int status) {
    switch (status) {
        case OFASR_OK: return "ok";
        case OFASR_ERR_INVALID_ARG: return "invalid argument";
        case OFASR_ERR_UNSUPPORTED: return "unsupported shape or dtype";
        case OFASR_ERR_WORKSPACE: return "workspace missing or too small";
        case OFASR_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}
