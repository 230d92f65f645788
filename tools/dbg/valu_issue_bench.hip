// Issue cost of the vector instructions the FPS inner loop is made of (one wave on a SIMD, independent streams):
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/valu_issue_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND>
__global__ void bench(unsigned long long *out, float seed) {
  f32x2 a[8]; float s[16];
  for (int i = 0; i < 8; ++i) a[i] = (f32x2){seed + i, seed * 2 + i};
  for (int i = 0; i < 16; ++i) s[i] = seed + i;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[i]));
      if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(a[i]));
      if (KIND == 2) { asm volatile("v_mul_f32 %0, %0, %0" : "+v"(s[2 * i])); asm volatile("v_mul_f32 %0, %0, %0" : "+v"(s[2 * i + 1])); }
      if (KIND == 3) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(s[2 * i]) : "v"(s[2 * i + 1])); asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(s[2 * i + 1]) : "v"(s[2 * i])); }
      if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (KIND == 5) { asm volatile("v_cmp_eq_f32 vcc, %0, %1\n\ts_cmp_eq_u64 vcc, 0\n\ts_cselect_b32 s20, 1, 0" :: "v"(s[2 * i]), "v"(s[2 * i + 1]) : "vcc", "s20", "scc"); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
  for (int i = 0; i < 16; ++i) acc += s[i];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)acc; }
}
int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  const char *names[] = {"8 x v_pk_mul_f32", "8 x v_pk_add_f32", "16 x v_mul_f32", "8 x (v_min_f32 + v_max3_f32)", "8 x v_pk_fma_f32", "8 x (v_cmp_eq_f32 -> s_cmp -> s_cselect)"};
  for (int k = 0; k < 6; ++k) {
    for (int rep = 0; rep < 2; ++rep) {
      if (k == 0) bench<0><<<1, 64>>>(d, 1.0f); if (k == 1) bench<1><<<1, 64>>>(d, 1.0f); if (k == 2) bench<2><<<1, 64>>>(d, 1.0f);
      if (k == 3) bench<3><<<1, 64>>>(d, 1.0f); if (k == 4) bench<4><<<1, 64>>>(d, 1.0f); if (k == 5) bench<5><<<1, 64>>>(d, 1.0f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s %7.1f cycles per group of 8 (one wave alone on its SIMD)\n", names[k], (double)h[0] / REP);
  }
  return 0;
}
