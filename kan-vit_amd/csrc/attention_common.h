// Shared between the attention translation units (attention.hip: tile forms 1-4 and the host entry points; attention16.hip:
// the 16-row-tile kernels of round 4).
#pragma once
#include "kanvit_common.h"

struct AttnArgs {
    const float* q;
    const float* k;
    const float* v;
    const float* o;
    const float* lse_in;
    const float* d_o;
    const float* delta_in;
    float* out;
    float* lse;
    float* dq;
    float* dk;
    float* dv;
    float* delta;
    float* ds;        // [B*H][NP][NP] dS = P*scale*(dP - delta), written by attn_bwd_kv2_kernel<.., DSOUT>, read by attn_bwd_dq_kernel
    long long qsb, qsh, qsn, ksb, ksh, ksn, vsb, vsh, vsn, osb, osh, osn;
    int B, H, N, D, causal, nkt, vec;
    int third;        // backward: the third-form fp32 kernels run (decided once in kanvit_attn_bwd)
    float scale;
};

constexpr int KV_N_CU = 256;           // MI355X
constexpr float KV_LOG2E = 1.4426950408889634f;

// 16-row-tile kernels (csrc/attention16.hip).  Each returns 1 when the launch is outside its domain (the caller continues with the
// older forms), 0 on success, < 0 on error.
int kv_attn16_fwd(const AttnArgs& a, hipStream_t st);
int kv_attn16_bwd(const AttnArgs& a, hipStream_t st);
int kv_attn16_fwd_bf16(const AttnArgs& a, hipStream_t st);      // KANVIT_FLAG_BF16_MFMA
int kv_attn16_bwd_bf16(const AttnArgs& a, hipStream_t st);
// the exact-fp32 backward of this shape runs kv_attn16_bwd and needs no dS hand-off in the workspace (host-side shape test only)
bool kv_attn16_bwd_ok(const kanvit_attn_desc* d);
