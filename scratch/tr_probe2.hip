#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int WG_T = 64, WG_STEP = 32, WG_LD = WG_T + 8;
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short *tile, int col0, int lane) {
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned short *a0 = tile + (8 * grp + q) * WG_LD + col0 + 4 * pp;
  const unsigned short *a1 = a0 + 4 * WG_LD;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);  // one whole-vector cast: element-wise casts of the tr-read result miscompile
}
__global__ __launch_bounds__(256) void k(const unsigned short* Gy, int ldg, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short s_g[WG_STEP * WG_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int sr = t >> 3, sc = (t & 7) * 8;
  uint4 vg = *(const uint4 *)(Gy + (long long)sr * ldg + sc);
  __syncthreads();
  *(uint4 *)&s_g[sr * WG_LD + sc] = vg;
  __syncthreads();
  bf16x8 f = tr_frag(s_g, (wave>>1)*32, lane);
  typedef __attribute__((ext_vector_type(8))) short s16x8; s16x8 fs = __builtin_bit_cast(s16x8, f);
  for (int j=0;j<8;j++) out[t*8+j] = (unsigned short)fs[j];
}
int main(){ unsigned short h[32*64]; for(int r=0;r<32;r++)for(int c=0;c<64;c++) h[r*64+c]=r*64+c; // raw bit patterns as ids
 unsigned short *d,*o; hipMalloc(&d,sizeof(h)); hipMalloc(&o,256*8*2); hipMemcpy(d,h,sizeof(h),hipMemcpyHostToDevice);
 k<<<1,256>>>(d,64,o); unsigned short ho[2048]; hipMemcpy(ho,o,4096,hipMemcpyDeviceToHost);
 for(int t: {0,1,15,16,17,63,64,128,130}){ printf("t %3d:",t); for(int j=0;j<8;j++) printf(" (%d,%d)", ho[t*8+j]/64, ho[t*8+j]%64); printf("\n"); } return 0; }
