// stubs.hip — C-ABI entry points whose kernels have not landed yet: they exist so that every symbol in
// include/sfgwas_hip.h resolves, and fail loudly when called.  Entries move out of this file as they are built.
#include "common.hpp"
#define NOT_YET(ctx, name) do { (ctx)->err = name ": not implemented in this build"; return 2; } while (0)
extern "C" {
int sfg_encode_coeffs_host(sfg_ctx *ctx, const double *, int, int64_t *) { NOT_YET(ctx, "sfg_encode_coeffs_host"); }
int sfg_beaver_elem_dev(sfg_ctx *ctx, int, int, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, uint64_t *, size_t) { NOT_YET(ctx, "sfg_beaver_elem_dev"); }
int sfg_beaver_elem(sfg_ctx *ctx, int, int, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, uint64_t *, size_t) { NOT_YET(ctx, "sfg_beaver_elem"); }
int sfg_beaver_matmul(sfg_ctx *ctx, int, int, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, uint64_t *, int, int, int) { NOT_YET(ctx, "sfg_beaver_matmul"); }
int sfg_sketch(sfg_ctx *ctx, const sfg_geno *, const int32_t *, const int8_t *, int, double *, uint64_t *, uint64_t *) { NOT_YET(ctx, "sfg_sketch"); }
int sfg_fill_uniform_ct_dev(sfg_ctx *ctx, uint64_t *, int, int, uint64_t) { NOT_YET(ctx, "sfg_fill_uniform_ct_dev"); }
int sfg_fill_geno_dev(sfg_ctx *ctx, int8_t *, size_t, size_t, uint64_t) { NOT_YET(ctx, "sfg_fill_geno_dev"); }
int sfg_fill_rotkeys_synthetic(sfg_ctx *ctx, const int *, int, uint64_t) { NOT_YET(ctx, "sfg_fill_rotkeys_synthetic"); }
}
