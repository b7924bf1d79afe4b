// Arguments of the batched per-step operators (rk_kalman_*_batched): shared by the lane-per-item kernels for n_state <= 16
// (kalman_batched.hip) and the workgroup-per-item kernel for larger blocks (solve_dense_ops.hpp).
#pragma once
#include "common.hpp"

namespace rk {

enum OpId { OP_PREDICT, OP_UPDATE, OP_FILTER, OP_SMOOTH_MV, OP_SMOOTH_SIM, OP_SMOOTH, OP_FORECAST, OP_SMOOTH_COND };

struct OpArgs {
    int n, p, m, op, sqrt_form;
    // inputs (NULL -> zeros)
    const double *mean_state_past, *var_state_past, *mean_state, *wgt_state, *var_state;
    const double *mean_state_pred, *var_state_pred, *x_meas, *mean_meas, *wgt_meas, *var_meas;
    const double *mean_state_next, *var_state_next, *mean_state_filt, *var_state_filt, *x_state_next;
    // outputs
    double *o_mean_pred, *o_var_pred, *o_mean_filt, *o_var_filt;
    double *o_mean_smooth, *o_var_smooth, *o_mean_sim, *o_var_sim;
    double *o_mean_fore, *o_var_fore, *o_wgt_cond, *o_mean_cond, *o_var_cond;
};


// n_state > 16: one 512-thread workgroup per item (solve_dense.hip)
int dense_op_launch(rk_handle h, const OpArgs& a);

}  // namespace rk
