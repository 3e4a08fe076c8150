// gpf_kernels.hpp -- gfx950 kernels of the particle-filter hot path.
//
// Layout in HBM (per handle / shard of N particles):
//   rows[2]  N x W Float64 "particle rows" (W = d or 2d, even), ping-pong    (Gen traces, flattened)
//   lw       N Float64 log-weights                                           (state.log_weights)
//   cdf[3]   N u64 inclusive fixed-point prefix sums (weights / residual counts / residual weights)
//   anc      N i32 ancestor of every output slot                              (state.parents)
// Rows instead of one array per column: the resample gather is random-access BY PARTICLE, and a
// 16..64-byte row is fetched with one or a few 16-byte lane loads from one cache line, where a
// column layout touches W different lines per particle.
//
// All kernels are wave64 / 256-thread workgroups; none uses MFMA (nothing here is a contraction).
#pragma once
#include "gpf_models.hpp"

#include "gpf_k_common.hpp"
#include "gpf_k_step.hpp"
#include "gpf_k_scan.hpp"
#include "gpf_k_search.hpp"
#include "gpf_k_fused.hpp"
#include "gpf_k_gather.hpp"
#include "gpf_k_sort.hpp"
#include "gpf_k_shard.hpp"
#include "gpf_k_resize.hpp"
#include "gpf_k_block.hpp"
