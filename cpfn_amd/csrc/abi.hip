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

// ---------------------------------------------------------------- cross-stream ordering by device flags
// The replayed training step is two linear graphs on two streams (the step; the next batch's geometry).  Ordering them
// with events — main: wait(geometry written), record(geometry read); side: wait(read), record(written) — costs ~100 us
// of a 1.9 ms step on this stack whatever the events' flags (each of the four operations alone is free; the closed
// cross-queue dependency cycle is not; hipStreamWaitValue32 / WriteValue32 are slower still), while a tiny eager kernel
// between two replays costs nothing.  So the streams order themselves: a one-lane kernel that polls a 4-byte flag until
// it has reached `value` (agent-scope acquire loads, s_sleep between them), and a one-lane kernel that stores a value
// (agent-scope release).  The data handed over is written by kernels that precede the setter on its stream and read by
// kernels that follow the waiter on its stream: kernel boundaries make it visible, exactly as with an event.
// A waiter gives up after `timeout_ticks` of the 100 MHz wall clock (the setter never came: a host-side error between the
// two launches, or the other stream stalled for longer than the caller allowed for), sets *err = 1 (pinned host word the
// host polls) AND *fault = 1.0f (device word: the trainer hands it to the optimizer as its "skip this step" flag, so that
// nothing computed after a broken hand-over can reach the weights — the host may already have queued several steps by the
// time it sees *err), and lets its stream continue instead of hanging the GPU.  Both words are sticky.
namespace {
__global__ void flag_wait_kernel(const unsigned *flag, unsigned value, unsigned long long timeout_ticks, unsigned *err,
                                 float *fault) {
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
    if ((unsigned long long)wall_clock64() - t0 > timeout_ticks) {
      if (fault) __hip_atomic_store(fault, 1.0f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (err) *(volatile unsigned *)err = 1u;       // (plain store: err may live in pinned host memory)
      break;
    }
    __builtin_amdgcn_s_sleep(16);
  }
}
__global__ void flag_set_kernel(unsigned *flag, unsigned value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
}
// flag_set with a small payload: up to 64 ints travel in the launch's arguments and are stored to `dst` BEFORE the flag (the
// trainer's FPS seeds of the next batch: they were a 128-byte host-to-device copy, i.e. a blit kernel of its own between
// two replays).  The waiter's stream sees them like everything else written before the flag.
namespace {
struct FlagPayload { int v[64]; };
__global__ void flag_set_payload_kernel(unsigned *flag, unsigned value, int *dst, FlagPayload p, int count) {
  if ((int)threadIdx.x < count) dst[threadIdx.x] = p.v[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
}
extern "C" int cpfn_flag_set_payload(unsigned *flag, unsigned value, int *dst, const int *payload, int count, void *stream) {
  if (!flag || count < 0 || count > 64 || (count > 0 && (!dst || !payload))) return -1;
  FlagPayload p;
  for (int i = 0; i < 64; ++i) p.v[i] = i < count ? payload[i] : 0;
  flag_set_payload_kernel<<<1, 64, 0, (hipStream_t)stream>>>(flag, value, dst, p, count);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int cpfn_flag_wait(const unsigned *flag, unsigned value, unsigned long long timeout_ticks, unsigned *err,
                              float *fault, void *stream) {
  if (!flag) return -1;
  flag_wait_kernel<<<1, 1, 0, (hipStream_t)stream>>>(flag, value, timeout_ticks, err, fault);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int cpfn_flag_set(unsigned *flag, unsigned value, void *stream) {
  if (!flag) return -1;
  flag_set_kernel<<<1, 1, 0, (hipStream_t)stream>>>(flag, value);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------- background geometry
// Set around geometry passes that run on a side stream BESIDE other work (the next batch's FPS / ball query / 3-NN beside
// a training step): those calls then pick the kernel shapes that disturb their neighbours least instead of the fastest
// ones (csrc/neighbors.hip, csrc/sampling.hip: measured both ways).
// Per THREAD since round 5 (as a process-global, two threads capturing at once got each other's setting): the
// switch is read at launch time by the thread that issues the launch, which is the thread that set it.
static thread_local bool g_background_geometry = false;
bool cpfn_background_geometry() { return g_background_geometry; }
extern "C" int cpfn_set_background_geometry(int on) {
  const int was = g_background_geometry ? 1 : 0;
  g_background_geometry = on != 0;
  return was;
}
