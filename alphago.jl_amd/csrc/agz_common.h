// agz_common.h -- shared host-side helpers of libagz (error plumbing, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/agz.h"
#include "../../include/agz_debug.h"

namespace agz {

struct Error : std::runtime_error {
  agz_status status;
  Error(agz_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};

inline std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return buf;
}

#define AGZ_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw ::agz::Error(AGZ_HIP_ERROR, ::agz::fmt("%s failed: %s (%s:%d)", #expr,            \
                                                   hipGetErrorString(_e), __FILE__, __LINE__)); \
  } while (0)

#define AGZ_REQUIRE(cond, status, ...)                                  \
  do {                                                                  \
    if (!(cond)) throw ::agz::Error((status), ::agz::fmt(__VA_ARGS__)); \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count == 0) return;
    AGZ_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void ensure(size_t count) {
    if (count > n) alloc(count);
  }
  void zero(hipStream_t s) {
    if (p) AGZ_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace agz
