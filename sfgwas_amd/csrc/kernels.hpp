// kernels.hpp — internal launch interface between the translation units of libsfgwas_hip.
#pragma once
#include "common.hpp"

// row r of a batch uses modulus m[r % period] (ciphertext rows, plaintext rows, key-switch rows are all periodic)
struct ModPattern { int period; int8_t m[32]; };

// ntt.hip
int launch_ntt_fwd(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat);
int launch_ntt_inv(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat);
int launch_ntt_plain(sfg_ctx *ctx, const long long *pc, u64 *out, size_t nplain, int L);
int launch_mac(sfg_ctx *ctx, const u64 *rot, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate);
