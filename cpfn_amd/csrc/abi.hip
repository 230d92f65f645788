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
