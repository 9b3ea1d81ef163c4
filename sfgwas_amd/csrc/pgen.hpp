// pgen.hpp — internal interface of pgen.hip (PLINK 2 .pgen index + device decode), shared with stream.hip
#pragma once
#include "common.hpp"
struct PgenIndex { uint32_t nv = 0, ns = 0; std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint8_t> vrt; };
struct PgenWindow { size_t start = 0, nr = 0, lead = 0; uint64_t f0 = 0, f1 = 0; std::vector<uint64_t> off; std::vector<uint32_t> ldb; };
size_t pgen_header_bytes(const uint8_t *first12);
int pgen_index(sfg_ctx *ctx, const uint8_t *f, size_t bytes, size_t file_bytes, PgenIndex &ix);
int pgen_window(sfg_ctx *ctx, const PgenIndex &ix, size_t file_bytes, size_t v0, size_t v1, PgenWindow &w);
size_t pgen_pitch(const PgenIndex &ix);
size_t pgen_desc_bytes(size_t nr);
int pgen_upload_desc(sfg_ctx *ctx, hipStream_t st, const PgenIndex &ix, const PgenWindow &w, uint8_t *desc_dev);
int launch_pgen_decode(sfg_ctx *ctx, hipStream_t st, const uint8_t *file_dev, const uint8_t *desc_dev, size_t nr, uint32_t ns, size_t pitch, uint8_t *rows_dev, const int **err_dev);
int pgen_decode_error(sfg_ctx *ctx, int herr);
