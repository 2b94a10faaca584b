// lm.hip -- placeholder so that every symbol of include/eao_fusion.h links; replaced by the HIP LM next.
#include "common.h"
extern "C" {
eao_status eao_pose_optimization(const eao_pose_problem*, eao_pose_result*) { eao::set_error("not built yet"); return EAO_ERR_INTERNAL; }
eao_status eao_local_ba(const eao_ba_problem*, const volatile uint8_t*, eao_ba_result*) { eao::set_error("not built yet"); return EAO_ERR_INTERNAL; }
eao_status eao_last_lm_trace(double*, double*, int32_t*, int32_t, int32_t* n) { if (n) *n = 0; return EAO_OK; }
eao_status eao_last_lm_timing(float*, int32_t*) { return EAO_ERR_INTERNAL; }
}
