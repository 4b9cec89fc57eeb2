// abd_eval_kernels.hpp -- the evaluation kernels (dense panels, observation lists) and the standalone fixed-order sum.
#pragma once

#include "abd_dense.hpp"
#include "abd_sparse.hpp"
#include "abd_obs.hpp"

#define ABD_FIN_THREADS 1024
__global__ __launch_bounds__(ABD_FIN_THREADS) void abd_finalize_kernel(const double* __restrict__ partials, int n_blocks,
                                                                       double* __restrict__ out, double tag) {
  __shared__ double sm[ABD_FIN_PARTS * ABD_NOUT];
  finalize_chain<ABD_FIN_THREADS>(partials + (int64_t)blockIdx.x * n_blocks * ABD_NOUT, n_blocks,
                                  out + blockIdx.x * ABD_NOUT, sm, threadIdx.x, tag);
}

