// Host <-> device copies of PAGEABLE memory through the library's own page-locked staging buffers.
//
// hipMemcpy / hipMemcpyAsync on pageable memory let the runtime register the caller's pages with the GPU (userptr) and keep
// the registration in a cache keyed by the host address.  When the caller frees the memory (large blocks are munmap'ed), the
// allocator later hands the same address range to somebody else, and a copy from THAT buffer finds the stale registration:
// the copy engine faults, and the HSA event thread aborts the process -- seen as a sporadic SIGABRT inside cuadmm_init (one
// run in five after the tail-solve ops had uploaded 32 MB numpy arrays; rocgdb: main thread in hipMemcpy under
// DevBuf::upload, abort() from libhsa-runtime64's queue-exception handler).  Copies staged through buffers this library owns
// never register caller memory.  Two 16 MB halves: the host memcpy of one chunk overlaps the DMA of the previous one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>

#include "common.h"
#include "device_util.h"

namespace cuadmm {

namespace {

constexpr size_t kHalf = (size_t)16 << 20;

struct Staging {
  std::mutex mu;
  char* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipStream_t st = nullptr;
  int ensure() {
    if (buf[0]) return CUADMM_OK;
    for (int i = 0; i < 2; ++i) {
      CUADMM_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&buf[i]), kHalf, hipHostMallocDefault));
      CUADMM_HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    CUADMM_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return CUADMM_OK;
  }
};

// One staging state PER DEVICE (stream, events and buffers belong to the device that was current when they were made: a copy
// for another device issued on this stream would not be covered by its events).  Process-wide, never freed (the runtime may
// already be gone at exit).
constexpr int kMaxDevices = 64;
Staging* staging(int& rc) {
  static Staging per_device[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    set_error("staging: no current HIP device (or more than %d devices)", kMaxDevices);
    rc = CUADMM_ERR_NO_DEVICE;
    return nullptr;
  }
  rc = CUADMM_OK;
  return &per_device[dev];
}

}  // namespace

// Earlier work that touches the destination (a memset, a kernel) must have finished: the copies run on the staging stream,
// which is ordered with nothing else.  `after`: the caller's stream; null = the whole device (setup paths).
static int drain(hipStream_t after) {
  if (after) CUADMM_HIP_TRY(hipStreamSynchronize(after));
  else CUADMM_HIP_TRY(hipDeviceSynchronize());
  return CUADMM_OK;
}

// dst (device) <- src (pageable host).  Blocks until the data is on the device.
int staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t after) {
  if (bytes == 0) return CUADMM_OK;
  { int rc0 = drain(after); if (rc0) return rc0; }          // before the lock: another thread's solver may be waiting for it
  int rc = CUADMM_OK;
  Staging* sp = staging(rc);
  if (!sp) return rc;
  Staging& s = *sp;
  std::lock_guard<std::mutex> lk(s.mu);
  if ((rc = s.ensure())) return rc;
  size_t done = 0;
  for (int i = 0; done < bytes; i ^= 1) {
    const size_t n = std::min(kHalf, bytes - done);
    CUADMM_HIP_TRY(hipEventSynchronize(s.ev[i]));                 // the DMA that last read this half has finished
    std::memcpy(s.buf[i], static_cast<const char*>(src) + done, n);
    CUADMM_HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst) + done, s.buf[i], n, hipMemcpyHostToDevice, s.st));
    CUADMM_HIP_TRY(hipEventRecord(s.ev[i], s.st));
    done += n;
  }
  CUADMM_HIP_TRY(hipStreamSynchronize(s.st));
  return CUADMM_OK;
}

// dst (pageable host) <- src (device); blocks
int staged_d2h(void* dst, const void* src, size_t bytes, hipStream_t after) {
  if (bytes == 0) return CUADMM_OK;
  { int rc0 = drain(after); if (rc0) return rc0; }          // the producers have finished (null: every stream of the device)
  int rc = CUADMM_OK;
  Staging* sp = staging(rc);
  if (!sp) return rc;
  Staging& s = *sp;
  std::lock_guard<std::mutex> lk(s.mu);
  if ((rc = s.ensure())) return rc;
  size_t done = 0, copied = 0;
  size_t len[2] = {0, 0};
  for (int i = 0; copied < bytes; i ^= 1) {
    if (len[i]) {                                                   // drain the half issued two rounds ago
      CUADMM_HIP_TRY(hipEventSynchronize(s.ev[i]));
      std::memcpy(static_cast<char*>(dst) + copied, s.buf[i], len[i]);
      copied += len[i];
      len[i] = 0;
    }
    if (done < bytes) {
      const size_t n = std::min(kHalf, bytes - done);
      CUADMM_HIP_TRY(hipMemcpyAsync(s.buf[i], static_cast<const char*>(src) + done, n, hipMemcpyDeviceToHost, s.st));
      CUADMM_HIP_TRY(hipEventRecord(s.ev[i], s.st));
      len[i] = n;
      done += n;
    }
  }
  return CUADMM_OK;
}

// rows x width_bytes from a pageable host matrix (row stride src_pitch) into a pitched device matrix
int staged_h2d_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, hipStream_t after) {
  if (rows == 0 || width_bytes == 0) return CUADMM_OK;
  { int rc0 = drain(after); if (rc0) return rc0; }
  int rc = CUADMM_OK;
  Staging* sp = staging(rc);
  if (!sp) return rc;
  Staging& s = *sp;
  std::lock_guard<std::mutex> lk(s.mu);
  if ((rc = s.ensure())) return rc;
  if (width_bytes > kHalf) { set_error("staged_h2d_2d: row of %zu bytes exceeds the staging buffer", width_bytes); return CUADMM_ERR_INVALID; }
  const size_t per = std::max<size_t>(1, kHalf / width_bytes);
  size_t r0 = 0;
  for (int i = 0; r0 < rows; i ^= 1) {
    const size_t nr = std::min(per, rows - r0);
    CUADMM_HIP_TRY(hipEventSynchronize(s.ev[i]));
    for (size_t r = 0; r < nr; ++r)
      std::memcpy(s.buf[i] + r * width_bytes, static_cast<const char*>(src) + (r0 + r) * src_pitch, width_bytes);
    CUADMM_HIP_TRY(hipMemcpy2DAsync(static_cast<char*>(dst) + r0 * dst_pitch, dst_pitch, s.buf[i], width_bytes, width_bytes, nr,
                                    hipMemcpyHostToDevice, s.st));
    CUADMM_HIP_TRY(hipEventRecord(s.ev[i], s.st));
    r0 += nr;
  }
  CUADMM_HIP_TRY(hipStreamSynchronize(s.st));
  return CUADMM_OK;
}

}  // namespace cuadmm
