// Optional per-launch timing inside the library (bench.py's roofline leg; see d2r_gemm_timer in include/d2r_hip.h).
// A scope brackets what is launched on `stream` between its construction and destruction with a pair of HIP events when the
// timer is armed, and records (family, flops, algorithmic bytes).  Families < 10000 are GEMM families (gemm.hip adds the kernel
// variant); 10001 = d2r_xattn_fwd_multi (one launch), 10002 = d2r_xattn_bwd_multi (query-side launch + the product launch).
#pragma once
#include <hip/hip_runtime.h>

struct D2RTimerScope {
  bool armed = false;
  int family = 0;
  double flops = 0.0, bytes = 0.0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st;
  D2RTimerScope(hipStream_t stream, int family, double flops, double bytes);
  ~D2RTimerScope();
  D2RTimerScope(const D2RTimerScope&) = delete;
  D2RTimerScope& operator=(const D2RTimerScope&) = delete;
};
