// The C ABI's per-thread error state (bnm_last_error): no HIP in this header, so the device-free part of the ABI
// (bnm_capi_model.cpp: model parsing / serialisation) compiles with a plain host compiler - tests/asan/ builds it with
// -fsanitize=address,undefined.
#pragma once
#include <string>
#include "../../include/bitnetmcu_hip.h"

namespace bnm_internal {

extern thread_local std::string g_err;      // bnm_last_error() of the calling thread (bnm_capi_model.cpp)

inline int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

}  // namespace bnm_internal
