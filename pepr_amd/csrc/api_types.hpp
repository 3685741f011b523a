// api_types.hpp -- the opaque handle types of include/peprml.h (shared by api.cpp and jackknife.cpp)
#pragma once
#include "engine.hpp"

#include <condition_variable>
#include <memory>

// one queued single-gene call (pml_score / pml_optimize / pml_search): concurrent callers are coalesced into
// ONE device batch by whichever caller finds no batch in flight (SURVEY 8b "Threading")
struct pml_request;
struct pml_ctx {
    pml::Ctx c;
    std::mutex qmu; std::condition_variable qcv; std::vector<pml_request *> queue; bool leader = false;
    long long coalesced_batches = 0, coalesced_requests = 0;
    std::vector<std::unique_ptr<pml::Ctx>> workers;      // contexts (streams) of the groups a search call is dealt over (api.cpp)
};
struct pml_batch { pml::Batch b; pml_ctx *owner; };
// the arenas the worker contexts keep for the next grouped search go back to the driver before any other call sizes its
// batches by free HBM (caller holds ctx->c.mu; no grouped call is running then)
inline void pml_drop_worker_caches(pml_ctx *ctx) {
    for (auto &w : ctx->workers) if (w->arena_cache) { hipSetDevice(w->device); hipFree(w->arena_cache); w->arena_cache = nullptr; w->arena_cache_bytes = 0; }
}

// Floating-point control state.  The host side of the engine (Gamma quantiles, Brent, NJ, Newick printing) runs on the
// CALLER's thread -- a JVM worker, a Python thread -- and must not inherit that thread's MXCSR (flush-to-zero / denormals-
// are-zero / rounding mode): a last-bit change in a Gamma rate changes a likelihood in the 16th digit, and the optimisers'
// discrete decisions (Brent brackets, dirty-branch flags) amplify that into visibly different -- equally valid -- optima.
// Every computing entry point of the C ABI runs under the default state (round to nearest, no FTZ / DAZ, exceptions masked)
// and restores the caller's on return.  pml_debug_fpenv reports the states callers came in with.
#include <xmmintrin.h>
struct pml_fpguard {
    unsigned saved;
    pml_fpguard() : saved(_mm_getcsr()) { note(saved); _mm_setcsr(0x1F80u); }
    ~pml_fpguard() { _mm_setcsr((saved & ~0x3Fu) | (_mm_getcsr() & 0x3Fu)); }     // keep the exception flags raised meanwhile
    static void note(unsigned v);
};
