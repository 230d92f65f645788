#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(short* out) {
  __shared__ short lds[64*72];
  for (int e = threadIdx.x; e < 64*72; e += 64) { int r = e/72, c = e%72; lds[e] = (short)(r*100 + c); }
  __syncthreads();
  int lane = threadIdx.x, grp = lane>>4, q = (lane&15)>>2, pp = lane&3;
  short* a0 = lds + (8*grp + q)*72 + 0 + 4*pp;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a0);
  for (int j=0;j<4;j++) out[lane*4+j] = v[j];
}
int main(){ short* d; hipMalloc(&d, 64*4*2); k<<<1,64>>>(d); short h[256]; hipMemcpy(h,d,512,hipMemcpyDeviceToHost);
 for(int l=0;l<64;l++){ printf("lane %2d:", l); for(int j=0;j<4;j++) printf(" %5d", h[l*4+j]); printf("\n"); } return 0; }
