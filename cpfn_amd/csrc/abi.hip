// Library identification (callable without a GPU).
#include "common.h"

extern "C" int cpfn_abi_version(void) { return CPFN_ABI_VERSION; }

extern "C" const char *cpfn_build_info(void) {
  return "libcpfn_hip gfx950 (CDNA4, wave64) built " __DATE__ " " __TIME__;
}
