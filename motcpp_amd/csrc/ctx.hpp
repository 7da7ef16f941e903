// The context object behind mot_ctx* (shared by the translation units that launch kernels on its stream).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct mot_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
};
