// Internal launchers of guidance.hip that take the public per-step coefficient struct.
#pragma once
#include "../../include/cgd_mi355x.h"
#include "common.h"

typedef cgd_step_coef StepCoef;

int cgd_launch_pmv_blend(cgd_ctx* ctx, const float* x, const float* out6, float* x0, float* mean, float* logvar, float* xin, int B,
                         int H, int W, const StepCoef& k, hipStream_t s);
int cgd_launch_guidance_combine(cgd_ctx* ctx, const float* gclip, const float* xin, const float* x0, float* gdir, float* seed6,
                                float* part, int B, int H, int W, const StepCoef& k, float tv_scale, float range_scale,
                                float sat_scale, hipStream_t s);
int cgd_launch_grad_finish(cgd_ctx* ctx, const float* gdir, const float* gunet, float* g, float* part, int B, int H, int W,
                           hipStream_t s);
int cgd_launch_scalars(cgd_ctx* ctx, const float* clip_part, int n_clip, const float* l_part, const float* g_part, int B, int H, int W,
                       int use_magnitude, float* scalars, hipStream_t s);
int cgd_launch_sample_update(cgd_ctx* ctx, const float* x, const float* x0, const float* mean, const float* logvar, const float* g,
                             const float* noise, const float* scalars, float* sample, float* x0_out, int B, int H, int W,
                             const StepCoef& k, int mode, hipStream_t s);
