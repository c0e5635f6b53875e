/* Prints the layout of the C-ABI structs as the C compiler sees them (tests/test_abi.py compares
 * it with the ctypes mirror in lightfm_amd/_native.py). */
#include <stddef.h>
#include <stdio.h>
#include "../include/lfm_hip.h"
#define F(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void)
{
    printf("lfm_csr %zu\nlfm_model %zu\nlfm_opts %zu\n", sizeof(lfm_csr), sizeof(lfm_model), sizeof(lfm_opts));
    F(lfm_csr, indices); F(lfm_csr, indptr); F(lfm_csr, data); F(lfm_csr, rows); F(lfm_csr, cols); F(lfm_csr, nnz);
    F(lfm_model, item_W); F(lfm_model, user_bM); F(lfm_model, n_item_feat); F(lfm_model, n_user_feat); F(lfm_model, d);
    F(lfm_model, adadelta); F(lfm_model, lr); F(lfm_model, rho); F(lfm_model, eps); F(lfm_model, max_sampled);
    F(lfm_model, item_scale); F(lfm_model, user_scale);
    F(lfm_opts, mode); F(lfm_opts, launches_per_epoch); F(lfm_opts, first_batch); F(lfm_opts, max_waves);
    F(lfm_opts, neg_log); F(lfm_opts, sampled_log); F(lfm_opts, counters); F(lfm_opts, kernel_ms);
    F(lfm_opts, update_mode); F(lfm_opts, feat_kernel); F(lfm_opts, warp_kernel); F(lfm_opts, debug);
    F(lfm_opts, phase_cycles); F(lfm_opts, tile_ng); F(lfm_opts, in_flight); F(lfm_opts, history);
    F(lfm_opts, ramp_k); F(lfm_opts, launches); F(lfm_opts, kernel_used); F(lfm_opts, shared_cap); F(lfm_opts, pos_begin); F(lfm_opts, pos_end);
    F(lfm_opts, streams_used); F(lfm_opts, tile_ahead); F(lfm_opts, plan_flags);
    return 0;
}
