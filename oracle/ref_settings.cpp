// oracle/_ref: the reference's util/settings.h is one of the few files that compiles in this image (it includes only the standard library). This driver (ours)
// includes it FROM WHERE IT LIES under /root/reference/src and hands out the values of its compile-time macros, so tests/test_constants_cpu.py can check the
// regex extraction of tests/golden/make_constants_ref.py against what a compiler sees. util/settings.cpp itself needs <boost/bind.hpp> (absent): its values are
// extracted as text. TEST INFRASTRUCTURE ONLY; built by oracle/Makefile when the reference is present, output in oracle/_ref/ (git-ignored).
#include "util/settings.h"
#include <cstring>

extern "C" int ref_settings_macro(const char* name, int* out) {
    static const struct { const char* name; int value; } t[] = {
        {"PYR_LEVELS", PYR_LEVELS}, {"patternNum", patternNum}, {"patternPadding", patternPadding},
        {"SOLVER_SVD", SOLVER_SVD}, {"SOLVER_ORTHOGONALIZE_SYSTEM", SOLVER_ORTHOGONALIZE_SYSTEM}, {"SOLVER_ORTHOGONALIZE_POINTMARG", SOLVER_ORTHOGONALIZE_POINTMARG},
        {"SOLVER_ORTHOGONALIZE_FULL", SOLVER_ORTHOGONALIZE_FULL}, {"SOLVER_SVD_CUT7", SOLVER_SVD_CUT7}, {"SOLVER_REMOVE_POSEPRIOR", SOLVER_REMOVE_POSEPRIOR},
        {"SOLVER_USE_GN", SOLVER_USE_GN}, {"SOLVER_FIX_LAMBDA", SOLVER_FIX_LAMBDA}, {"SOLVER_ORTHOGONALIZE_X", SOLVER_ORTHOGONALIZE_X},
        {"SOLVER_MOMENTUM", SOLVER_MOMENTUM}, {"SOLVER_STEPMOMENTUM", SOLVER_STEPMOMENTUM}, {"SOLVER_ORTHOGONALIZE_X_LATER", SOLVER_ORTHOGONALIZE_X_LATER},
    };
    for (const auto& e : t) if (std::strcmp(e.name, name) == 0) { *out = e.value; return 1; }
    return 0;
}
