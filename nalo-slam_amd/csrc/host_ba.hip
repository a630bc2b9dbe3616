// placeholder: BA window (filled in next milestone)
#include "nalo_internal.h"
namespace nalo { struct BAWindow { int W = 0; }; void ba_destroy(nalo_ctx* c) { delete c->ba; c->ba = nullptr; } }
#define STUB(name, ...) int name(__VA_ARGS__) { return NALO_ERR_UNSUPPORTED; }
extern "C" {
STUB(nalo_ba_set_window, nalo_ctx*, int, const nalo_frame_state*, const double*, const double*)
STUB(nalo_ba_set_points, nalo_ctx*, int, const int*, const float*, const float*, const float*, const float*, const float*, const float*, const int*)
STUB(nalo_ba_set_residuals, nalo_ctx*, const uint8_t*)
STUB(nalo_ba_set_prior, nalo_ctx*, const double*, const double*)
STUB(nalo_ba_get_prior, nalo_ctx*, double*, double*)
STUB(nalo_ba_linearize, nalo_ctx*, int, double*)
STUB(nalo_ba_accumulate, nalo_ctx*, int, double*, double*)
STUB(nalo_ba_accumulate_sc, nalo_ctx*, int, double*, double*)
STUB(nalo_ba_solve_system, nalo_ctx*, int, double, double*)
STUB(nalo_ba_backup_state, nalo_ctx*)
STUB(nalo_ba_do_step, nalo_ctx*, float, float, float, float, float, int*)
STUB(nalo_ba_optimize, nalo_ctx*, int, int, double*)
STUB(nalo_ba_marginalize_points, nalo_ctx*, const uint8_t*, double*, double*, double*, double*)
STUB(nalo_ba_get_frames, nalo_ctx*, nalo_frame_state*, double*, double*)
STUB(nalo_ba_get_points, nalo_ctx*, float*, float*, float*, float*, float*, float*, float*, float*)
STUB(nalo_ba_get_residuals, nalo_ctx*, int8_t*, uint8_t*, float*, float*, float*)
STUB(nalo_ba_get_acc13, nalo_ctx*, double*)
STUB(nalo_ba_counts, nalo_ctx*, int*, int*, int*)
STUB(nalo_ba_set_allreduce, nalo_ctx*, nalo_allreduce_fn, void*)
STUB(nalo_dense_make_map, nalo_ctx*, int, const float*, float, const double*, int, int*, int*, int*, float*, float*, uint8_t*, int*, int*)
}
