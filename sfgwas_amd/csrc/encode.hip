// encode.hip — placeholder until the encoder lands (defined so the context builds)
#include "common.hpp"
int sfg_encoder_init(sfg_ctx *ctx) { return 0; }
void sfg_encoder_destroy(sfg_ctx *ctx) {}
