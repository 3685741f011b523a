// api_types.hpp -- the opaque handle types of include/peprml.h (shared by api.cpp and jackknife.cpp)
#pragma once
#include "engine.hpp"

struct pml_ctx { pml::Ctx c; };
struct pml_batch { pml::Batch b; pml_ctx *owner; };
