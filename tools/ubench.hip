// Instruction-throughput microbenchmark for gfx950 (design input for the ring-MAC / NTT kernels).
// Each kernel runs ITER iterations of UNROLL independent dependency chains of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
constexpr int ITER=4096;
typedef unsigned long long u64;

#define KERNEL8(name, decl, init, body, sink) \
__global__ void __launch_bounds__(256) name(u64* out, u64 seed){ \
  decl; init; \
  for(int it=0; it<ITER; ++it){ body } \
  sink; }

// fp64 FMA: 8 chains
__global__ void __launch_bounds__(256) k_fma_f64(u64* out, u64 seed){
  double a[8]; double b=(double)(seed|1)*1e-9, c=1.0000001;
  for(int i=0;i<8;i++) a[i]=(double)(threadIdx.x+i);
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_fma_f64 %0, %0, %1, %2":"+v"(a[i]):"v"(c),"v"(b));
  }
  double s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123.456) out[0]=(u64)s;
}
__global__ void __launch_bounds__(256) k_mul_f64(u64* out, u64 seed){
  double a[8]; double c=1.0000001;
  for(int i=0;i<8;i++) a[i]=(double)(threadIdx.x+i+1);
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mul_f64 %0, %0, %1":"+v"(a[i]):"v"(c));
  }
  double s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123.456) out[0]=(u64)s;
}
__global__ void __launch_bounds__(256) k_add_f64(u64* out, u64 seed){
  double a[8]; double c=1.0000001;
  for(int i=0;i<8;i++) a[i]=(double)(threadIdx.x+i+1);
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_add_f64 %0, %0, %1":"+v"(a[i]):"v"(c));
  }
  double s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123.456) out[0]=(u64)s;
}
__global__ void __launch_bounds__(256) k_rndne_f64(u64* out, u64 seed){
  double a[8];
  for(int i=0;i<8;i++) a[i]=(double)(threadIdx.x+i+1)*1.37;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_rndne_f64 %0, %0":"+v"(a[i]));
  }
  double s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123.456) out[0]=(u64)s;
}
__global__ void __launch_bounds__(256) k_mad_u64_u32(u64* out, u64 seed){
  u64 a[8]; unsigned x=(unsigned)seed|1, y=(unsigned)(seed>>7)|3;
  for(int i=0;i<8;i++) a[i]=threadIdx.x+i;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0":"+v"(a[i]):"v"(x),"v"(y):"vcc");
  }
  u64 s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_mul_lo_u32(u64* out, u64 seed){
  unsigned a[8]; unsigned y=(unsigned)(seed>>7)|3;
  for(int i=0;i<8;i++) a[i]=threadIdx.x+i+1;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mul_lo_u32 %0, %0, %1":"+v"(a[i]):"v"(y));
  }
  unsigned s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_mul_hi_u32(u64* out, u64 seed){
  unsigned a[8]; unsigned y=(unsigned)(seed>>7)|0xF0000003u;
  for(int i=0;i<8;i++) a[i]=0xFFFFFFF0u-threadIdx.x-i;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mul_hi_u32 %0, %0, %1":"+v"(a[i]):"v"(y));
  }
  unsigned s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_mad_u32_u24(u64* out, u64 seed){
  unsigned a[8]; unsigned y=(unsigned)(seed>>7)|3, x=(unsigned)seed|5;
  for(int i=0;i<8;i++) a[i]=threadIdx.x+i+1;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mad_u32_u24 %0, %1, %2, %0":"+v"(a[i]):"v"(x),"v"(y));
  }
  unsigned s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_mul_hi_u32_u24(u64* out, u64 seed){
  unsigned a[8]; unsigned y=(unsigned)(seed>>7)|0xF00003u;
  for(int i=0;i<8;i++) a[i]=0xFFFFF0u-threadIdx.x-i;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_mul_hi_u32_u24 %0, %0, %1":"+v"(a[i]):"v"(y));
  }
  unsigned s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_dot4_u32_u8(u64* out, u64 seed){
  unsigned a[8]; unsigned y=(unsigned)(seed>>7)|0x01020304u, x=(unsigned)seed|0x05060708u;
  for(int i=0;i<8;i++) a[i]=threadIdx.x+i+1;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0":"+v"(a[i]):"v"(x),"v"(y));
  }
  unsigned s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_add_u32(u64* out, u64 seed){
  unsigned a[8]; unsigned y=(unsigned)(seed>>7)|3;
  for(int i=0;i<8;i++) a[i]=threadIdx.x+i+1;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_add_u32 %0, %0, %1":"+v"(a[i]):"v"(y));
  }
  unsigned s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_addc(u64* out, u64 seed){
  unsigned a[8], b[8]; unsigned y=(unsigned)(seed>>7)|0xF0000003u;
  for(int i=0;i<8;i++){ a[i]=threadIdx.x+i+1; b[i]=i; }
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc":"+v"(a[i]),"+v"(b[i]):"v"(y):"vcc");
  }
  unsigned s=0; for(int i=0;i<8;i++) s+=a[i]+b[i]; if(s==123456) out[0]=s;
}
__global__ void __launch_bounds__(256) k_fma_f32(u64* out, u64 seed){
  float a[8]; float b=(float)(seed|1)*1e-9f, c=1.0000001f;
  for(int i=0;i<8;i++) a[i]=(float)(threadIdx.x+i);
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_fma_f32 %0, %0, %1, %2":"+v"(a[i]):"v"(c),"v"(b));
  }
  float s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123.456f) out[0]=(u64)s;
}
__global__ void __launch_bounds__(256) k_cvt_f64_u32(u64* out, u64 seed){
  double a[8]; unsigned x[8];
  for(int i=0;i<8;i++) x[i]=threadIdx.x+i+(unsigned)seed;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_cvt_f64_u32 %0, %1":"=v"(a[i]):"v"(x[i]));
  }
  double s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123.456) out[0]=(u64)s;
}
__global__ void __launch_bounds__(256) k_lshl_b64(u64* out, u64 seed){
  u64 a[8]; unsigned sh=(unsigned)(seed&1);
  for(int i=0;i<8;i++) a[i]=threadIdx.x+i+1;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++) asm volatile("v_lshlrev_b64 %0, %1, %0":"+v"(a[i]):"v"(sh));
  }
  u64 s=0; for(int i=0;i<8;i++) s+=a[i]; if(s==123456) out[0]=s;
}
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_mfma_f64(u64* out, u64 seed){
  d4 acc[4]; for(int i=0;i<4;i++) acc[i]=(d4){0,0,0,0};
  double a=(double)threadIdx.x, b=(double)(seed&7);
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<4;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<4;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3]; if(s==123.456) out[0]=(u64)s;
}
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_mfma_i8(u64* out, u64 seed){
  i16v acc[2]; for(int i=0;i<2;i++) for(int j=0;j<16;j++) acc[i][j]=0;
  i4v a={(int)threadIdx.x,1,2,3}, b={(int)(seed&7),5,6,7};
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<2;i++) acc[i]=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,acc[i],0,0,0);
  }
  int s=0; for(int i=0;i<2;i++) for(int j=0;j<16;j++) s+=acc[i][j]; if(s==123456) out[0]=(u64)s;
}
// mixed: MFMA f64 and VALU fma f64 in the same wave
__global__ void __launch_bounds__(256) k_mix_mfma_valu(u64* out, u64 seed){
  d4 acc[4]; for(int i=0;i<4;i++) acc[i]=(d4){0,0,0,0};
  double a=(double)threadIdx.x, b=(double)(seed&7);
  double v[8]; double c=1.0000001; for(int i=0;i<8;i++) v[i]=(double)(threadIdx.x+i);
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<4;i++){ acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
      asm volatile("v_fma_f64 %0, %0, %1, %2":"+v"(v[2*i]):"v"(c),"v"(b));
      asm volatile("v_fma_f64 %0, %0, %1, %2":"+v"(v[2*i+1]):"v"(c),"v"(b)); }
  }
  double s=0; for(int i=0;i<4;i++) s+=acc[i][0]+acc[i][1]; for(int i=0;i<8;i++) s+=v[i]; if(s==123.456) out[0]=(u64)s;
}
// LDS read rates
__global__ void __launch_bounds__(256) k_lds_b64(u64* out, u64 seed){
  __shared__ u64 lds[8192];
  for(int i=threadIdx.x;i<8192;i+=256) lds[i]=i+seed;
  __syncthreads();
  u64 s=0; unsigned base=threadIdx.x&63;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++){ s+=lds[base+64*((it+i)&127)]; }
  }
  if(s==123456) out[0]=s;
}
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_lds_b128(u64* out, u64 seed){
  __shared__ u64x2 lds[4096];
  for(int i=threadIdx.x;i<4096;i+=256) lds[i]=(u64x2){(u64)i+seed,(u64)i};
  __syncthreads();
  u64 s=0; unsigned base=threadIdx.x&63;
  for(int it=0;it<ITER;++it){
#pragma unroll
    for(int i=0;i<8;i++){ u64x2 v=lds[base+64*((it+i)&63)]; s+=v.x+v.y; }
  }
  if(s==123456) out[0]=s;
}

struct Case{ const char* name; void(*fn)(u64*,u64); double ops_per_thread; };
int main(){
  u64* d; CK(hipMalloc(&d,1024));
  std::vector<Case> cases={
    {"v_fma_f64",(void(*)(u64*,u64))k_fma_f64,8.0*ITER},{"v_mul_f64",(void(*)(u64*,u64))k_mul_f64,8.0*ITER},
    {"v_add_f64",(void(*)(u64*,u64))k_add_f64,8.0*ITER},{"v_rndne_f64",(void(*)(u64*,u64))k_rndne_f64,8.0*ITER},
    {"v_mad_u64_u32",(void(*)(u64*,u64))k_mad_u64_u32,8.0*ITER},{"v_mul_lo_u32",(void(*)(u64*,u64))k_mul_lo_u32,8.0*ITER},
    {"v_mul_hi_u32",(void(*)(u64*,u64))k_mul_hi_u32,8.0*ITER},{"v_mad_u32_u24",(void(*)(u64*,u64))k_mad_u32_u24,8.0*ITER},
    {"v_mul_hi_u32_u24",(void(*)(u64*,u64))k_mul_hi_u32_u24,8.0*ITER},{"v_dot4_u32_u8",(void(*)(u64*,u64))k_dot4_u32_u8,8.0*ITER},
    {"v_add_u32",(void(*)(u64*,u64))k_add_u32,8.0*ITER},{"add_co+addc pair",(void(*)(u64*,u64))k_addc,8.0*ITER},
    {"v_fma_f32",(void(*)(u64*,u64))k_fma_f32,8.0*ITER},{"v_cvt_f64_u32",(void(*)(u64*,u64))k_cvt_f64_u32,8.0*ITER},
    {"v_lshlrev_b64",(void(*)(u64*,u64))k_lshl_b64,8.0*ITER},
    {"mfma_f64_16x16x4 (per wave-instr)",(void(*)(u64*,u64))k_mfma_f64,4.0*ITER},
    {"mfma_i32_32x32x32_i8 (per wave-instr)",(void(*)(u64*,u64))k_mfma_i8,2.0*ITER},
    {"mix 1 mfma_f64 + 2 v_fma_f64 (per triple)",(void(*)(u64*,u64))k_mix_mfma_valu,4.0*ITER},
    {"ds_read_b64",(void(*)(u64*,u64))k_lds_b64,8.0*ITER},{"ds_read_b128",(void(*)(u64*,u64))k_lds_b128,8.0*ITER},
  };
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int wpc : {4,8,16}){           // waves per CU via blocks of 256 threads: blocks/CU = wpc/4
    int blocks=256*wpc/4;
    printf("== %d waves/CU (%d blocks x 256 thr)\n",wpc,blocks);
    for(auto&c:cases){
      hipLaunchKernelGGL(c.fn,dim3(blocks),dim3(256),0,0,d,(u64)0x1234567);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for(int r=0;r<5;r++) hipLaunchKernelGGL(c.fn,dim3(blocks),dim3(256),0,0,d,(u64)0x1234567);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=5;
      double total=c.ops_per_thread*256.0*blocks;   // lane-ops
      double rate=total/(ms*1e-3);
      // lanes per clk per SIMD at 2.4 GHz: rate/(1024 SIMDs*2.4e9)
      printf("%-44s %8.3f ms  %10.3e lane-ops/s  %6.2f lanes/clk/SIMD@2.4GHz\n",c.name,ms,rate,rate/(1024*2.4e9));
    }
  }
  return 0;
}
