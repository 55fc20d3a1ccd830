// core.hip — version, thread-local error string, launch checking.
#include <stdarg.h>

#include <atomic>
#include <mutex>

#include "common.h"

thread_local char d2r_err_buf[512] = "";

int d2r_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(d2r_err_buf, sizeof(d2r_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

int d2r_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return D2R_OK;
}

// Stream fork / join for the calls that issue independent parts of their work on caller-provided auxiliary streams
// (interaction.hip): `to` waits for everything enqueued on `from` so far.  Events come from a ring created on first use (host
// objects, timing disabled); a wait captures the event's state at the time of the call, so recording the same event again later
// does not disturb earlier waits.
static int event_from_ring(hipEvent_t* out) {
  constexpr int RING = 256;
  static hipEvent_t ring[RING];
  static std::atomic<unsigned> next{0};
  static std::once_flag once;
  static hipError_t init_err = hipSuccess;
  std::call_once(once, [] {
    for (int i = 0; i < RING && init_err == hipSuccess; ++i) init_err = hipEventCreateWithFlags(&ring[i], hipEventDisableTiming);
  });
  if (init_err != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "stream fork: hipEventCreate failed: %s", hipGetErrorString(init_err));
  *out = ring[next.fetch_add(1) % RING];
  return D2R_OK;
}
// the two halves of a fork, for a wait that has to land later in the waiting stream's queue than the point of the record
int d2r_event_record(void* stream, void** ev) {
  hipEvent_t e;
  if (int rc = event_from_ring(&e)) return rc;
  hipError_t err = hipEventRecord(e, (hipStream_t)stream);
  if (err != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "d2r_event_record: %s", hipGetErrorString(err));
  *ev = (void*)e;
  return D2R_OK;
}
int d2r_stream_wait(void* stream, void* ev) {
  hipError_t err = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
  if (err != hipSuccess) return d2r_fail(D2R_ERR_LAUNCH, "d2r_stream_wait: %s", hipGetErrorString(err));
  return D2R_OK;
}
int d2r_stream_fork(void* from, void* to) {
  if (from == to) return D2R_OK;
  void* ev = nullptr;
  if (int rc = d2r_event_record(from, &ev)) return rc;
  return d2r_stream_wait(to, ev);
}

extern "C" const char* d2r_version(void) { return "d2r_hip 0.1.0 (gfx950)"; }
extern "C" const char* d2r_last_error(void) { return d2r_err_buf; }
