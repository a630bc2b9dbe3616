// Every constant of the reference that the kernels and the host code of this path consume, in ONE table. The table defines the
// constexpr values the code uses (nalo::k...) AND is what nalo_constants() dumps, so the dump is the set of values the kernels were compiled with.
// tests/test_constants_gpu.py / tests/test_constants_cpu.py compare the dump with tests/golden/constants_ref.json, which tests/golden/make_constants_ref.py
// extracts from the reference's own sources (paths relative to src/):
//   util/settings.cpp:56-160,297   util/settings.h:52,232-234   FullSystem/HessianBlocks.h:61-68,268   util/NumType.h:41-53
// X(type, C++ name, reference name, value)
#pragma once

#define NALO_REF_CONSTANTS(X)                                                                       \
    /* FullSystem/HessianBlocks.h:61-68 */                                                           \
    X(float, kScaleIdepth, "SCALE_IDEPTH", 1.0f)                                                     \
    X(float, kScaleXiRot, "SCALE_XI_ROT", 1.0f)                                                      \
    X(float, kScaleXiTrans, "SCALE_XI_TRANS", 0.5f)                                                  \
    X(float, kScaleF, "SCALE_F", 50.0f)                                                              \
    X(float, kScaleC, "SCALE_C", 50.0f)                                                              \
    X(float, kScaleA, "SCALE_A", 10.0f)                                                              \
    X(float, kScaleB, "SCALE_B", 1000.0f)                                                            \
    /* util/settings.h:52,232-234; util/NumType.h:41-53 */                                           \
    X(int, kPyrLevels, "PYR_LEVELS", 6)                                                              \
    X(int, kPatternNum, "patternNum", 8)                                                             \
    X(int, kPatternPadding, "patternPadding", 2)                                                     \
    X(int, kMaxResPerPoint, "MAX_RES_PER_POINT", 8)                                                  \
    X(int, kNumThreads, "NUM_THREADS", 6)                                                            \
    X(int, kCPars, "CPARS", 4)                                                                       \
    /* util/settings.cpp */                                                                          \
    X(float, kIdepthFixPrior, "setting_idepthFixPrior", 50.0f * 50.0f)                               \
    X(float, kIdepthFixPriorMargFac, "setting_idepthFixPriorMargFac", 600.0f * 600.0f)               \
    X(double, kInitialRotPrior, "setting_initialRotPrior", 1e11f)                                    \
    X(double, kInitialTransPrior, "setting_initialTransPrior", 1e10f)                                \
    X(double, kInitialAffBPrior, "setting_initialAffBPrior", 1e14f)                                  \
    X(double, kInitialAffAPrior, "setting_initialAffAPrior", 1e14f)                                  \
    X(double, kInitialCalibHessian, "setting_initialCalibHessian", 5e9f)                             \
    X(double, kSolverModeDelta, "setting_solverModeDelta", 0.00001)                                  \
    X(int, kForceAcceptStep, "setting_forceAceptStep", 1)                                            \
    X(float, kMinIdepthHAct, "setting_minIdepthH_act", 100.0f)                                       \
    X(float, kMinIdepthHMarg, "setting_minIdepthH_marg", 50.0f)                                      \
    X(int, kMinFrames, "setting_minFrames", 5)                                                       \
    X(int, kMaxFrames, "setting_maxFrames", 7)                                                       \
    X(int, kMaxOptIterations, "setting_maxOptIterations", 6)                                         \
    X(int, kMinOptIterations, "setting_minOptIterations", 1)                                         \
    X(float, kThOptIterations, "setting_thOptIterations", 1.2f)                                      \
    X(float, kOutlierTH, "setting_outlierTH", 12.0f * 12.0f)                                         \
    X(float, kOutlierTHSumComponent, "setting_outlierTHSumComponent", 50.0f * 50.0f)                 \
    X(double, kMargWeightFac, "setting_margWeightFac", 0.5f * 0.5f)                                  \
    X(int, kPhotometricCalibration, "setting_photometricCalibration", 2)                             \
    X(double, kAffineOptModeA, "setting_affineOptModeA", 1e12f)                                      \
    X(double, kAffineOptModeB, "setting_affineOptModeB", 1e8f)                                       \
    X(float, kHuberTH, "setting_huberTH", 9.0f)                                                      \
    X(float, kFrameEnergyTHConstWeight, "setting_frameEnergyTHConstWeight", 0.5f)                    \
    X(float, kFrameEnergyTHN, "setting_frameEnergyTHN", 0.7f)                                        \
    X(float, kFrameEnergyTHFacMedian, "setting_frameEnergyTHFacMedian", 1.5f)                        \
    X(float, kOverallEnergyTHWeight, "setting_overallEnergyTHWeight", 1.0f)                          \
    X(float, kCoarseCutoffTH, "setting_coarseCutoffTH", 20.0f)                                       \
    X(float, kMinGradHistCut, "setting_minGradHistCut", 0.5f)                                        \
    X(float, kMinGradHistAdd, "setting_minGradHistAdd", 7.0f)                                        \
    X(float, kGradDownweightPerLevel, "setting_gradDownweightPerLevel", 0.75f)                       \
    X(float, kImmMaxPixSearch, "setting_maxPixSearch", 0.027f)                                       \
    X(float, kMinTraceQuality, "setting_minTraceQuality", 3.0f)                                      \
    X(int, kImmMinTraceTestRadius, "setting_minTraceTestRadius", 2)                                  \
    X(int, kImmGNItsActivation, "setting_GNItsOnPointActivation", 3)                                 \
    X(float, kImmStepsize, "setting_trace_stepsize", 1.0f)                                           \
    X(int, kImmGNIts, "setting_trace_GNIterations", 3)                                               \
    X(float, kImmGNTh, "setting_trace_GNThreshold", 0.1f)                                            \
    X(float, kImmExtraSlack, "setting_trace_extraSlackOnTH", 1.2f)                                   \
    X(float, kImmSlackInterval, "setting_trace_slackInterval", 1.5f)                                 \
    X(float, kImmMinImprovement, "setting_trace_minImprovementFactor", 2.0f)                         \
    X(int, kSparsityFactorInit, "sparsityFactor", 5)                                                 \
    /* FullSystem/HessianBlocks.h:268: frameEnergyTH = 8*8*patternNum */                             \
    X(float, kFrameEnergyTHInit, "frameEnergyTH_init", 8 * 8 * 8)

namespace nalo {
#define NALO_REF_DEFINE(type, name, refname, value) constexpr type name = value;
NALO_REF_CONSTANTS(NALO_REF_DEFINE)
#undef NALO_REF_DEFINE
// the eight pattern offsets (staticPattern[8], util/settings.cpp:297; "8 for SSE efficiency"), x then y; host and device code index these two arrays
#define NALO_PATTERN_DX {0, -1, 1, -2, 0, 2, -1, 0}
#define NALO_PATTERN_DY {-2, -1, -1, 0, 0, 0, 1, 2}
constexpr int kPatternDx[8] = NALO_PATTERN_DX, kPatternDy[8] = NALO_PATTERN_DY;
constexpr double kInitialAffPrior = kInitialAffAPrior;        // A and B carry the same value in the reference; the code that treats them alike uses this name
static_assert(kInitialAffAPrior == kInitialAffBPrior, "the frame-0 prior code assumes one affine prior");
}  // namespace nalo
