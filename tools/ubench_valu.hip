// Micro-benchmark: VALU integer throughput on gfx950 (issue cycles per wave64 instruction).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITERS = 4096;
constexpr uint32_t P = 0x78000001u, MU = 0x88000001u;

template<int OP> __device__ __forceinline__ void step(uint32_t& a, uint32_t b) {
  if constexpr (OP==0) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==1) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==2) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==3) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==4) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==5) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==6) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==7) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==8) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  else if constexpr (OP==9) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); }
}

template<int OP> __global__ void k_op(uint32_t* out, uint32_t seed) {
  uint32_t a[8];
  uint32_t b = seed + threadIdx.x;
  #pragma unroll
  for (int i=0;i<8;i++) a[i] = seed*(i+1) + threadIdx.x;
  for (int it=0; it<ITERS; it++) {
    #pragma unroll
    for (int i=0;i<8;i++) step<OP>(a[i], b);
  }
  uint32_t s=0;
  #pragma unroll
  for (int i=0;i<8;i++) s ^= a[i];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}

// 64-bit mad
__global__ void k_mad64(uint32_t* out, uint32_t seed) {
  uint64_t a[8]; uint32_t b = seed + threadIdx.x;
  #pragma unroll
  for (int i=0;i<8;i++) a[i] = seed*(i+1) + threadIdx.x;
  for (int it=0; it<ITERS; it++) {
    #pragma unroll
    for (int i=0;i<8;i++) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"((uint32_t)b), "v"((uint32_t)(seed+i)) : "vcc"); }
  }
  uint64_t s=0;
  #pragma unroll
  for (int i=0;i<8;i++) s ^= a[i];
  out[blockIdx.x*blockDim.x+threadIdx.x] = (uint32_t)s ^ (uint32_t)(s>>32);
}

__global__ void k_fma64(uint32_t* out, uint32_t seed) {
  double a[8]; double b = 1.0 + 1e-9*(seed + threadIdx.x);
  #pragma unroll
  for (int i=0;i<8;i++) a[i] = seed*(i+1) + threadIdx.x;
  for (int it=0; it<ITERS; it++) {
    #pragma unroll
    for (int i=0;i<8;i++) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b)); }
  }
  double s=0;
  #pragma unroll
  for (int i=0;i<8;i++) s += a[i];
  out[blockIdx.x*blockDim.x+threadIdx.x] = (uint32_t)s;
}

// Montgomery mul variants (C-level, compiler-scheduled)
__device__ __forceinline__ uint32_t monty_mul_v0(uint32_t a, uint32_t b) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * MU;
  uint32_t u = (uint32_t)(((uint64_t)m * P) >> 32);
  uint32_t hi = (uint32_t)(t >> 32);
  uint32_t r = hi - u;
  return hi < u ? r + P : r;
}
// variant: m via shifts (MU = 2^31 + 2^27 + 1)
__device__ __forceinline__ uint32_t monty_mul_v1(uint32_t a, uint32_t b) {
  uint64_t t = (uint64_t)a * b;
  uint32_t lo = (uint32_t)t;
  uint32_t m = lo + (lo << 27) + (lo << 31);
  uint32_t u = __umulhi(m, P);
  uint32_t hi = (uint32_t)(t >> 32);
  uint32_t r = hi - u;
  return min(r, r + P);   // if hi<u, r wrapped (huge) and r+P is the right (small) value; else r<P<=r+P... r+P may not overflow: r<P so r+P<2^32 ok
}
// variant: positive form: r = (t + m'*P)>>32, m' = lo * (-P^-1)
__device__ __forceinline__ uint32_t monty_mul_v2(uint32_t a, uint32_t b) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * (0u - MU);
  uint64_t s = t + (uint64_t)m * P;      // mad_u64_u32
  uint32_t r = (uint32_t)(s >> 32);
  return min(r, r - P);
}
// variant: 16-bit limb split with 24-bit multiplies
__device__ __forceinline__ uint32_t monty_mul_v3(uint32_t a, uint32_t b) {
  // full 62-bit product from 4 u24 muls
  uint32_t a0 = a & 0xffff, a1 = a >> 16, b0 = b & 0xffff, b1 = b >> 16;
  uint32_t p00 = __umul24(a0,b0), p01 = __umul24(a0,b1), p10 = __umul24(a1,b0), p11 = __umul24(a1,b1);
  uint64_t t = (uint64_t)p00 + (((uint64_t)p01 + p10) << 16) + ((uint64_t)p11 << 32);
  uint32_t lo = (uint32_t)t;
  uint32_t m = lo + (lo << 27) + (lo << 31);
  uint32_t u = __umulhi(m, P);
  uint32_t hi = (uint32_t)(t >> 32);
  uint32_t r = hi - u;
  return min(r, r + P);
}
template<int V> __global__ void k_monty(uint32_t* out, uint32_t seed) {
  uint32_t a[8]; uint32_t b = (seed*77 + threadIdx.x) % P;
  #pragma unroll
  for (int i=0;i<8;i++) a[i] = (seed*(i+1) + threadIdx.x) % P;
  for (int it=0; it<ITERS; it++) {
    #pragma unroll
    for (int i=0;i<8;i++) {
      if constexpr (V==0) a[i] = monty_mul_v0(a[i], b);
      else if constexpr (V==1) a[i] = monty_mul_v1(a[i], b);
      else if constexpr (V==2) a[i] = monty_mul_v2(a[i], b);
      else a[i] = monty_mul_v3(a[i], b);
    }
  }
  uint32_t s=0;
  #pragma unroll
  for (int i=0;i<8;i++) s ^= a[i];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}

template<typename F> int run(const char* name, F launch, int blocks, int threads, double ops_per_thread) {
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r=0;r<5;r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=5;
  double total_lane_ops = ops_per_thread * blocks * (double)threads;
  double wave_instr = total_lane_ops/64.0;
  // cycles per wave-instr per SIMD at 2.4GHz, 1024 SIMDs
  double cyc = ms*1e-3*2.4e9*1024.0 / wave_instr;
  printf("%-22s %8.3f ms  %8.2f Tlaneops/s  %6.2f cyc/wave-op/SIMD(@2.4GHz)\n", name, ms, total_lane_ops/ms*1e-9, cyc);
  return 0;
}

int main() {
  int blocks = 256*8*4, threads = 256;   // 8 waves/SIMD resident, 4 rounds
  uint32_t* out; CK(hipMalloc(&out, (size_t)blocks*threads*4));
  double n = (double)ITERS*8;
  #define RUN_OP(OP,NAME) run(NAME, [&]{ hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, blocks, threads, n)
  RUN_OP(0,"v_mul_lo_u32"); RUN_OP(1,"v_mul_hi_u32"); RUN_OP(2,"v_mul_u32_u24"); RUN_OP(3,"v_mad_u32_u24");
  RUN_OP(6,"v_mul_hi_u32_u24"); RUN_OP(4,"v_add_u32"); RUN_OP(5,"v_lshl_add_u32"); RUN_OP(7,"v_min_u32"); RUN_OP(8,"v_sub_u32"); RUN_OP(9,"v_xor_b32");
  run("v_mad_u64_u32", [&]{ hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, blocks, threads, n);
  run("v_fma_f64", [&]{ hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, blocks, threads, n);
  run("monty_mul v0 (3 mul)", [&]{ hipLaunchKernelGGL(k_monty<0>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, blocks, threads, n);
  run("monty_mul v1 (shift m)", [&]{ hipLaunchKernelGGL(k_monty<1>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, blocks, threads, n);
  run("monty_mul v2 (positive)", [&]{ hipLaunchKernelGGL(k_monty<2>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, blocks, threads, n);
  run("monty_mul v3 (u24 limbs)", [&]{ hipLaunchKernelGGL(k_monty<3>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, blocks, threads, n);
  // simple copy bandwidth check
  return 0;
}
