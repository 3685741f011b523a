// api_types.hpp -- the opaque handle types of include/peprml.h (shared by api.cpp and jackknife.cpp)
#pragma once
#include "engine.hpp"

#include <condition_variable>

// one queued single-gene call (pml_score / pml_optimize / pml_search): concurrent callers are coalesced into
// ONE device batch by whichever caller finds no batch in flight (SURVEY 8b "Threading")
struct pml_request;
struct pml_ctx {
    pml::Ctx c;
    std::mutex qmu; std::condition_variable qcv; std::vector<pml_request *> queue; bool leader = false;
    long long coalesced_batches = 0, coalesced_requests = 0;
};
struct pml_batch { pml::Batch b; pml_ctx *owner; };
