// Does the VGPR bank of an fp64 instruction's source operands matter on gfx950?  Eight independent chains of one instruction with explicit physical registers:
// the chain registers are v[16+4i : 17+4i] ("0": pair starts on a register = 0 mod 4) or v[18+4i : 19+4i] ("2"), the shared operands v[4:5], v[6:7], v[8:9].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
constexpr int ITER = 4096;
#define CLOB "v4","v5","v6","v7","v8","v9","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49"
#define INIT "v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_mov_b32 v6, 1\n v_mov_b32 v7, 0x3ff00000\n v_mov_b32 v8, 2\n v_mov_b32 v9, 0x3e700000\n" \
  "v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n" \
  "v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0x3ff00000\n" \
  "v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3ff00000\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0x3ff00000\n" \
  "v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0x3ff00000\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3ff00000\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0x3ff00000\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0x3ff00000\n"
// chain i uses register pair base B + 4 i
#define CH8(OP, B) OP(B) OP(B+4) OP(B+8) OP(B+12) OP(B+16) OP(B+20) OP(B+24) OP(B+28)
#define STR2(x) #x
#define STR(x) STR2(x)
#define KERNEL(name, BODY) __global__ void __launch_bounds__(256) name(double *out) { asm volatile(INIT ::: CLOB); for (int it = 0; it < ITER; ++it) asm volatile(BODY ::: CLOB); double r; asm volatile("v_add_f64 %0, v[16:17], v[18:19]" : "=v"(r) :: CLOB); if (r == 123.456) out[0] = r; }
// fma d = a * b + d
#define FMA_46(B)  "v_fma_f64 v[" STR(B) ":" STR(B+1) "], v[4:5], v[6:7], v[" STR(B) ":" STR(B+1) "]\n"
#define FMA_48(B)  "v_fma_f64 v[" STR(B) ":" STR(B+1) "], v[4:5], v[8:9], v[" STR(B) ":" STR(B+1) "]\n"
#define FMA_44(B)  "v_fma_f64 v[" STR(B) ":" STR(B+1) "], v[4:5], v[4:5], v[" STR(B) ":" STR(B+1) "]\n"
#define ADD_4(B)   "v_add_f64 v[" STR(B) ":" STR(B+1) "], v[4:5], v[" STR(B) ":" STR(B+1) "]\n"
#define ADD_6(B)   "v_add_f64 v[" STR(B) ":" STR(B+1) "], v[6:7], v[" STR(B) ":" STR(B+1) "]\n"
#define MUL_4(B)   "v_mul_f64 v[" STR(B) ":" STR(B+1) "], v[4:5], v[" STR(B) ":" STR(B+1) "]\n"
#define MUL_6(B)   "v_mul_f64 v[" STR(B) ":" STR(B+1) "], v[6:7], v[" STR(B) ":" STR(B+1) "]\n"
// three DISTINCT chain-dependent registers: d = x_i * y + d with x_i another chain's register (read only)
#define FMA_X0(B)  "v_fma_f64 v[" STR(B) ":" STR(B+1) "], v[" STR(B+2) ":" STR(B+3) "], v[4:5], v[" STR(B) ":" STR(B+1) "]\n"
KERNEL(k_fma_a4b6_d0, CH8(FMA_46, 16))      // a bank 0, b bank 2, d bank 0
KERNEL(k_fma_a4b6_d2, CH8(FMA_46, 18))      // a 0, b 2, d 2
KERNEL(k_fma_a4b8_d0, CH8(FMA_48, 16))      // a 0, b 0, d 0: all three in one bank pair
KERNEL(k_fma_a4b8_d2, CH8(FMA_48, 18))      // a 0, b 0, d 2
KERNEL(k_fma_a4a4_d2, CH8(FMA_44, 18))      // a = b (one read), d 2
KERNEL(k_fma_x_d0, CH8(FMA_X0, 16))         // a = v[B+2] (bank 2), b 0, d 0
KERNEL(k_add_4_d0, CH8(ADD_4, 16))          // both bank 0
KERNEL(k_add_6_d0, CH8(ADD_6, 16))          // 2 and 0
KERNEL(k_mul_4_d0, CH8(MUL_4, 16))
KERNEL(k_mul_6_d0, CH8(MUL_6, 16))
// straight-line code: the same eight chains unrolled to 512 / 4096 instructions (4 / 32 KiB of VOP3 code) inside the loop - does instruction fetch keep up?
#define REP8(X) X X X X X X X X
#define KERNEL_N(name, BODY, DIV) __global__ void __launch_bounds__(256) name(double *out) { asm volatile(INIT ::: CLOB); for (int it = 0; it < ITER / DIV; ++it) asm volatile(BODY ::: CLOB); double r; asm volatile("v_add_f64 %0, v[16:17], v[18:19]" : "=v"(r) :: CLOB); if (r == 123.456) out[0] = r; }
KERNEL_N(k_fma_code4k, REP8(REP8(CH8(FMA_46, 16))), 64)
KERNEL_N(k_fma_code32k, REP8(REP8(REP8(CH8(FMA_46, 16)))), 512)
KERNEL_N(k_mix_code32k, REP8(REP8(REP8(CH8(FMA_46, 16)))) REP8(REP8(REP8(CH8(ADD_6, 16)))), 1024)
int main() {
    double *out; CK(hipMalloc(&out, 64));
    struct { const char *n; void (*f)(double *); } ks[] = {{"fma a0 b2 d0", k_fma_a4b6_d0}, {"fma a0 b2 d2", k_fma_a4b6_d2}, {"fma a0 b0 d0 (all one bank pair)", k_fma_a4b8_d0}, {"fma a0 b0 d2", k_fma_a4b8_d2},
        {"fma a0 a0 d2 (a = b)", k_fma_a4a4_d2}, {"fma x2 b0 d0", k_fma_x_d0}, {"add 0 0", k_add_4_d0}, {"add 2 0", k_add_6_d0}, {"mul 0 0", k_mul_4_d0}, {"mul 2 0", k_mul_6_d0},
        {"fma, 4 KiB of straight-line code", k_fma_code4k}, {"fma, 32 KiB of straight-line code", k_fma_code32k}, {"fma+add, 64 KiB of straight-line code", k_mix_code32k}};
    for (int blocks : {512, 1024, 256}) {
        printf("== %d waves per CU\n", blocks * 4 / 256);
        for (auto &k : ks) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out);
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = (double)blocks * 256 * ITER * 8;
            printf("%-36s %.3f ms  %6.2f lanes/clk/SIMD@2.4GHz\n", k.n, ms, ops / (ms * 1e-3) / (256.0 * 4 * 2.4e9));
        }
    }
    return 0;
}
