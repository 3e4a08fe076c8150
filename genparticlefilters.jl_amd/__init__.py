"""MI355X-native particle-filter hot path behind the pf_initialize / pf_update! / pf_resample! /
pf_rejuvenate! API of probcomp/GenParticleFilters.jl.  The directory name contains a
dot, so import it through the repo-root shim:  `import gpf_amd`."""
from . import _lib, models                                        # noqa: F401
from .api import *                                                # noqa: F401,F403
from .api import (DeviceParticleFilterState, DeviceParticleFilterSubState, ParticleFilterState, ParticleFilterSubState, ParticleFilterView, ErrorException, Tempering, mh, move_reweight, locally_optimal, line_fixed, MoveProposal, locally_optimal_move, outlier_propose,
                  pf_initialize, pf_update, pf_step_ess, choiceproduct, pf_resample, pf_multinomial_resample, pf_residual_resample,
                  pf_stratified_resample, pf_resample_blocks, block_resampled, block_stats, pf_initialize_blocks, pf_update_blocks, pf_rejuvenate_blocks, pf_rejuvenate, pf_move_accept, pf_move_reweight,
                  pf_resize, pf_multinomial_resize, pf_residual_resize, pf_optimal_resize, pf_replicate, pf_dereplicate,
                  effective_sample_size, get_ess, log_ml_estimate, get_lml_est, get_log_weights,
                  get_log_norm_weights, get_norm_weights, get_traces, sample_unweighted_traces, mean, var, proportionmap)
