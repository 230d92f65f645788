// fp32 matrices -> bf16 (or fp32) panels with a row stride, up to MCV_MAX of them per launch: the per-step refresh of every
// bf16 weight panel of the network.  Shared by cpfn_multi_cast (gather.hip) and by the launch that runs it beside sa1's fp32
// first layer (bn.hip: cpfn_smallk_fwd_cast).
#pragma once
#include "common.h"

constexpr int MCV_MAX = 64;
struct McvArgs {
  const float *src[MCV_MAX];
  void *dst[MCV_MAX];
  int rows[MCV_MAX], cols[MCV_MAX], ld[MCV_MAX], f32[MCV_MAX];
  int sld[MCV_MAX];            // row stride of the source (== cols: contiguous; > cols: a column slice of a wider matrix)
  int block0[MCV_MAX + 1];
  int count;
};
// (bx: the workgroup's index among the cast workgroups; NT threads per workgroup)
template <int NT>
__device__ __forceinline__ void multi_cast_body(const McvArgs &a, int bx) {
  int d = 0;
  while (d + 1 < a.count && bx >= a.block0[d + 1]) ++d;
  const int cols = a.cols[d], ld = a.ld[d], sld = a.sld[d];
  const long long n = (long long)a.rows[d] * cols;
  const long long nb = a.block0[d + 1] - a.block0[d];
  const float *__restrict__ s = a.src[d];
  for (long long e = ((long long)(bx - a.block0[d]) * NT + threadIdx.x) * 4; e < n; e += nb * NT * 4) {
    float v[4];
    const long long r = e / cols;
    int c = (int)(e - r * cols);
    if (sld != cols) {                       // column slice: element by element
      long long rr = r;
      int cc = c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = e + j < n ? s[rr * sld + cc] : 0.f;
        if (++cc == cols) { cc = 0; ++rr; }
      }
    } else if (e + 4 <= n && (((uintptr_t)(s + e)) & 15) == 0) {
      const float4 q = *(const float4 *)(s + e);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = e + j < n ? s[e + j] : 0.f;
    }
    long long o = r * ld + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (e + j < n) {
        if (a.f32[d]) ((float *)a.dst[d])[o] = v[j];
        else ((unsigned short *)a.dst[d])[o] = __builtin_bit_cast(unsigned short, (__bf16)v[j]);
      }
      ++o;
      if (++c == cols) { c = 0; o += ld - cols; }
    }
  }
}

// host: descriptors [0, count) (count <= MCV_MAX) -> kernel arguments + number of workgroups; CPFN_EINVAL on a bad descriptor
static inline int mcv_fill(const cpfn_cast_desc *descs, int count, McvArgs &a, int *blocks_out) {
  a.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const cpfn_cast_desc &d = descs[i];
    if (!d.src || !d.dst || d.rows < 0 || d.cols <= 0 || d.dst_ld < d.cols || ((uintptr_t)d.src & 3) ||
        (d.src_ld != 0 && d.src_ld < d.cols)) return CPFN_EINVAL;
    a.src[i] = d.src; a.dst[i] = d.dst; a.rows[i] = d.rows; a.cols[i] = d.cols; a.ld[i] = d.dst_ld; a.f32[i] = d.dst_f32;
    a.sld[i] = d.src_ld > 0 ? d.src_ld : d.cols;
    a.block0[i] = blocks;
    long long nb = ((long long)d.rows * d.cols + 256 * 4 - 1) / (256 * 4);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    blocks += (int)nb;
  }
  a.block0[count] = blocks;
  *blocks_out = blocks;
  return 0;
}
