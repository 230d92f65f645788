// Library identification (callable without a GPU).
#include "common.h"

extern "C" int cpfn_abi_version(void) { return CPFN_ABI_VERSION; }

extern "C" const char *cpfn_build_info(void) {
  return "libcpfn_hip gfx950 (CDNA4, wave64) built " __DATE__ " " __TIME__;
}

// Rate of the constant device wall clock (wall_clock64 / s_memrealtime) in kHz; <= 0 on error.
extern "C" int cpfn_wall_clock_khz(int device) {
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess) return -1;
  return khz;
}

// One device wall-clock reading (100 MHz) into *dst, as a kernel on `stream`: a time stamp INSIDE a captured graph
// (CPFN_STEP_STAMPS=1 in training.py: when did the geometry branch / the main chain of the replayed step end?).
namespace {
__global__ void stamp_kernel(unsigned long long *dst) { *dst = (unsigned long long)wall_clock64(); }
}
extern "C" int cpfn_stamp(unsigned long long *dst, void *stream) {
  if (!dst) return -1;
  stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(dst);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
