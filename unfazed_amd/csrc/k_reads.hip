// k_reads.hip -- read stage (placeholder until the kernels land)
#include "uz_ctx.hpp"
void uz_launch_phase(uz_ctx *c, FamilyDev &f, const SitesDev &s, ReadsDev &r, int32_t *status, int32_t *counts,
                     int32_t *origin, int32_t *evidence) {
    throw UzError{UZ_E_STATE, "read stage not built"};
}
int uz_phase_votes_impl(uz_ctx *c, int64_t *vote_off, int32_t *vote_val) { return UZ_E_STATE; }
int uz_phase_groups_impl(uz_ctx *c, int64_t *grp_off, int32_t *grp_q) { return UZ_E_STATE; }
void uz_phase_state_free(uz_ctx *c) {}
